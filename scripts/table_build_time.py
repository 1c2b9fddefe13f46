"""Time of the per-call proposal-table build (alpha_choice_kernel through the irec_test_proposal_table hook) and of small calls
as issued / with the tables kept; diagnostics only (GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import bench, irec
eng = irec.get_engine()


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for r in range(reps):
        fn(); ev[r + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(reps))
    return ts[0], ts[len(ts) // 2]


for D, S, steps in ((1000, 36, 32), (192, 36, 32), (1000, 20, 32), (56, 20, 32), (1000, 36, 8), (1000, 148, 32)):
    mn, med = timed(lambda: eng.test_proposal_table(42, S, D, steps))
    print(f"table D={D} S={S} steps={steps}: min {mn * 1e3:.1f} us, median {med * 1e3:.1f} us (hook: includes the output allocation)", flush=True)

for L, n, bs, B, omega, eps1 in ((38, 8192, 1000, 20, 3.0, 1.2), (28, 8192, 1000, 20, 3.0, 1.2), (1, 8192, 1000, 20, 3.0, 1.2),
                                 (1, 301056, 1000, 10, 3.0, 1.0), (1, 12288, 1000, 10, 3.0, 1.0), (256, 8192, 1000, 20, 3.0, 1.2)):
    S = int(np.exp(omega * eps1))
    from oracle import oracle as O
    st = [O.synthetic_latent(1234 + i, n) for i in range(L)]
    q = [torch.from_numpy(np.stack([s[k] for s in st])).cuda().contiguous() for k in range(4)]
    lay = eng.layout(L, n, bs, 42)
    for name, flags in (("as issued", 0), ("tables kept", irec._lib.IREC_FLAG_REUSE_TABLES)):
        params = eng.params(omega, S, B, flags)
        mn, med = timed(lambda: eng.encode_blocks(params, lay, *q, 42, 48))
        print(f"{lay.n_blocks} blocks B={B} S={S} {eng.plan(params, lay, 48)['kernel']} {name}: min {mn:.4f} ms, median {med:.4f} ms", flush=True)
