import torch, time
dev = torch.device("cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
n = int(2.4e6)
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
def serial():
    torch.cuda._sleep(n); torch.cuda._sleep(n)
def par():
    with torch.cuda.stream(s1): torch.cuda._sleep(n)
    with torch.cuda.stream(s2): torch.cuda._sleep(n)
for _ in range(2): serial(); par()
print("eager: serial %.2f ms, two streams %.2f ms" % (t(serial), t(par)))
# graphs
x1 = torch.zeros(1, device=dev); x2 = torch.zeros(1, device=dev)
def body(x):
    for _ in range(50):
        torch.cuda._sleep(n // 50); x.add_(1)
gs = []
for st, x in ((s1, x1), (s2, x2)):
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st): body(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st): body(x)
    gs.append(g)
def g_serial():
    gs[0].replay(); gs[1].replay()
def g_par():
    cur = torch.cuda.current_stream()
    for st, g in zip((s1, s2), gs):
        st.wait_stream(cur)
        with torch.cuda.stream(st): g.replay()
    for st in (s1, s2): cur.wait_stream(st)
for _ in range(2): g_serial(); g_par()
print("graphs (100 nodes each): same stream %.2f ms, two streams %.2f ms" % (t(g_serial), t(g_par)))
