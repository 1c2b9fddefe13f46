"""Times one encode variant (IREC_VARIANT = table | fused | generic) on the bench workload; diagnostics only."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import bench, irec
variant = os.environ.get("IREC_VARIANT", "table")
L = int(os.environ.get("LATENTS", "1024"))
B = int(os.environ.get("BEAMS", "20")); omega = float(os.environ.get("OMEGA", "3.0")); eps1 = float(os.environ.get("EPS1", "1.2"))
eng = irec.get_engine()
S = int(np.exp(omega * eps1))
flags = {"table": 8, "auto": 0, "one_table": 4, "fused": 2, "generic": 1}[variant]
flags |= irec._lib.IREC_FLAG_SHAPE[os.environ.get("SHAPE", "default")]   # team-encoder workgroup shape (diagnostics)
flags |= int(os.environ.get("SPLIT_W", "0")) << 12                         # split-encoder width (diagnostics)
if os.environ.get("NO_SPLIT"): flags |= 16
if os.environ.get("NO_TEN"): flags |= irec._lib.IREC_FLAG_NO_TEN            # at most ten beams: encode_team_kernel<10,..> instead of encode_ten_kernel
max_K = int(os.environ.get("MAXK", "32"))
params = eng.params(omega, S, B, flags, table_steps=int(os.environ.get("TABLE_STEPS", "0")))
# SKEW=1: per-tensor log-normal scale on delta (bench.skewed_batch): K from 1 to 60 inside one call
q = bench.skewed_batch(L, eng.device, 0) if os.environ.get("SKEW") else bench.synthetic_batch(L, eng.device, 0)
by_K = bool(os.environ.get("ORDER_BY_K"))
lay = eng.layout(L, bench.N_DIMS, bench.BLOCK_SIZE, bench.SEED)
order = os.environ.get("ORDER", "")   # diagnostic: the order in which the persistent kernel meets the blocks
if order:
    dim = lay.block_dim.cpu().numpy()
    nat = np.argsort(lay.order, kind="stable")                    # rows in natural (tensor, block) order
    if order == "natural":
        rows = nat
    elif order == "natural_tail":                                 # natural, but the last small blocks close the launch
        small = nat[dim[nat] < dim.max()]
        keep_tail = set(small[-int(os.environ.get("TAIL", "1536")):].tolist())
        rows = np.array([r for r in nat if r not in keep_tail] + [r for r in nat if r in keep_tail])
    elif order == "spread":                                       # big blocks first-come, one small block after every 8 big
        big, small = np.nonzero(dim == dim.max())[0], np.nonzero(dim < dim.max())[0]
        tail = int(os.environ.get("TAIL", "1536"))
        body_small = small[:-tail] if tail < len(small) else small[:0]
        per = max(1, len(big) // max(1, len(body_small)))
        rows = []
        js = 0
        for i, r in enumerate(big):
            rows.append(r)
            if (i + 1) % per == 0 and js < len(body_small):
                rows.append(body_small[js]); js += 1
        rows += body_small[js:].tolist() + small[len(body_small):].tolist()
        rows = np.array(rows)
    assert sorted(rows.tolist()) == list(range(lay.n_blocks))
    lay = lay.subset(rows)
out = None
print("plan:", eng.plan(params, lay, max_K)["kernel"], flush=True)
for i in range(int(os.environ.get("REPS", "3"))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = eng.encode_blocks(params, lay, *q, bench.SEED, max_K, order_by_K=by_K)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{variant} B={B} S={S}: {dt * 1e3:.2f} ms for {L} latents -> {L / dt:.0f} latents/s", flush=True)
