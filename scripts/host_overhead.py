"""Host side of a mid-size call: how long the CPU takes to issue one irec_beam_encode (Python mirror -> ctypes -> planning -> two launches)
against what the GPU takes to run it; diagnostics only.  CASE = kodak1 | share342 | nine | one20 (one tensor at the headline settings) | ten (34 latents at the reference's default settings); DIMS / TENSORS override the sizes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import bench, irec
eng = irec.get_engine()
dev = eng.device
case = os.environ.get("CASE", "kodak1")
omega, eps1, B, nt, nd = {"kodak1": (3.0, 1.0, 10, 1, 301056), "share342": (3.0, 1.2, 20, 38, 8192), "nine": (3.0, 1.2, 20, 1, 8192), "one20": (3.0, 1.2, 20, 1, 301056), "ten": (3.0, 1.0, 10, 34, 8192)}[case]
nd = int(os.environ.get("DIMS", nd)); nt = int(os.environ.get("TENSORS", nt))
S = int(np.exp(omega * eps1)); max_K = 32
flags = irec._lib.IREC_FLAG_REUSE_TABLES if os.environ.get("KEEP") else 0
if os.environ.get("LISTED"): flags |= irec._lib.IREC_FLAG_LISTED_ORDER   # rows dealt as listed, not by cost
params = eng.params(omega, S, B, flags)
q = bench.skewed_batch(nt, dev, 0, nd, sigma=float(os.environ["SKEW"])) if os.environ.get("SKEW") else bench.synthetic_batch(nt, dev, 77, nd)   # SKEW = sigma of a per-tensor log-normal scale on delta: K differs between tensors
lay = eng.layout(nt, nd, bench.BLOCK_SIZE, bench.SEED)
out = (torch.empty(lay.n_blocks, dtype=torch.int32, device=dev), torch.empty((lay.n_blocks, max_K), dtype=torch.int32, device=dev), torch.empty_like(q[0]))
print("plan:", eng.plan(params, lay, max_K)["kernel"], flush=True)
for _ in range(5): eng.encode_blocks(params, lay, *q, bench.SEED, max_K, out=out)
torch.cuda.synchronize()
N = int(os.environ.get("REPS", "200"))
for trial in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(N): eng.encode_blocks(params, lay, *q, bench.SEED, max_K, out=out)
    e1.record(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{case}: host issue {1e6 * (t1 - t0) / N:.1f} us/call; GPU {1e3 * e0.elapsed_time(e1) / N:.1f} us/call back to back; wall {1e6 * (t2 - t0) / N:.1f} us/call", flush=True)
# with an event between the calls, as bench.py times them
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record()
for r in range(N):
    eng.encode_blocks(params, lay, *q, bench.SEED, max_K, out=out); ev[r + 1].record()
torch.cuda.synchronize()
d = [ev[r].elapsed_time(ev[r + 1]) * 1e3 for r in range(N)]
print(f"{case}: event to event median {np.median(d):.1f} us, min {np.min(d):.1f}", flush=True)
