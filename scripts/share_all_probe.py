"""Probe (GPU box): calls of 64 ... 252 blocks -- the default (one 8-wave team per CU) against every row shared between the teams
of the two-team build (IREC_FLAG_SHARE_ALL + shape 2), tables kept and as issued."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import irec
from oracle import oracle as O
eng = irec.get_engine()
F = irec._lib


def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for r in range(reps):
        fn(); ev[r + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(reps))
    return ts[0], ts[len(ts) // 2]


n, bs, B, omega, eps1 = 8192, 1000, int(os.environ.get("BEAMS", "20")), 3.0, float(os.environ.get("EPS1", "1.2"))
S = int(np.exp(omega * eps1))
for L in (8, 10, 12, 14, 16, 18, 20, 24, 28):
    st = [O.synthetic_latent(1234 + i, n) for i in range(L)]
    q = [torch.from_numpy(np.stack([s[k] for s in st])).cuda().contiguous() for k in range(4)]
    lay = eng.layout(L, n, bs, 42)
    line = f"{lay.n_blocks:4d} blocks B={B}:"
    for name, fl in (("default", 0), ("share-all, two teams", F.IREC_FLAG_SHARE_ALL | F.IREC_FLAG_SHAPE["2"]), ("share-all, three teams", F.IREC_FLAG_SHARE_ALL)):
        for kept in (1, 0):
            params = eng.params(omega, S, B, fl | (F.IREC_FLAG_REUSE_TABLES if kept else 0))
            plan = eng.plan(params, lay, 32)
            mn, med = timed(lambda: eng.encode_blocks(params, lay, *q, 42, 32))
            line += f" | {name} ({plan['kernel'].split('kernel')[1]} W={plan['split']}) {'kept' if kept else 'as issued'} {mn:.4f}"
    print(line, flush=True)
