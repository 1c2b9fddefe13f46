"""Diagnostic: latency of calls of FEW blocks of more than 1024 dims -- the reference's default block_size=None on one image's latents --
coded by gangs of teams (irec_team.hip, "Gangs") against every block on one team (IREC_FLAG_NO_SPLIT).
Usage: python scripts/gang_latency.py [--huge] [--stripes] | python scripts/with_lib.py gang_ablN scripts/gang_latency.py --ablate   (--huge: also ONE block of 301 056 dims, Kodak level 1 whole: ~2 200 partitions)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import bench, irec
from irec import _lib
eng = irec.get_engine()


def run(label, n_t, n, bs, B, S, max_K, reps, flags=0, table_steps=None):
    table_steps = max_K if table_steps is None else table_steps   # (a window over every partition, as BeamSearchCoder sizes it from the K it has seen;
                                                                  #  the library bounds the bytes: steps beyond the window draw their rows in the kernel)
    q = bench.synthetic_batch(n_t, eng.device, 0) if n == bench.N_DIMS else None
    if q is None:
        from oracle import oracle as O
        st = [O.synthetic_latent(500 + i, n) for i in range(n_t)]
        q = tuple(torch.from_numpy(np.stack([s[k] for s in st])).to(eng.device).contiguous() for k in range(4))
    lay = eng.layout(n_t, n, bs, bench.SEED)
    out = {}
    for name, fl in (("gang", flags), ("one team", flags | _lib.IREC_FLAG_NO_SPLIT)):
        params = eng.params(3.0, S, B, fl, table_steps=table_steps)
        plan = eng.plan(params, lay, max_K)
        K, idx, s = eng.encode_blocks(params, lay, *q, bench.SEED, max_K)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            K, idx, s = eng.encode_blocks(params, lay, *q, bench.SEED, max_K)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        Kh = K.cpu().numpy().astype(np.int64); dims = lay.block_dim.cpu().numpy().astype(np.int64)
        assert Kh.min() >= 0 or "--ablate" in sys.argv, Kh
        evals = float((S * dims * (1 + np.maximum(Kh - 1, 0) * B) * (Kh > 0)).sum())
        out[name] = (K, idx, s)
        print(f"{label:34s} {name:8s} {plan['kernel']:34s} grid {plan['grid']:3d} x{plan['teams_per_wg']} split {plan['split']:3d}: {dt * 1e3:10.3f} ms, "
              f"max K {int(Kh.max()):4d}, {evals / dt / (plan['n_cu'] * plan['clock_mhz'] * 1e6):6.3f} look-ups/clk/CU (chip), "
              f"{dt / max(1, int(Kh.max())) * 1e6:7.1f} us/step", flush=True)
    if "--ablate" in sys.argv:      # (a diagnostic build with phases removed: wrong outputs, only the time counts)
        return
    a, b = out["gang"], out["one team"]
    Kh = a[0].cpu().numpy(); ia, ib = a[1].cpu().numpy(), b[1].cpu().numpy()
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and all(np.array_equal(ia[r, :Kh[r]], ib[r, :Kh[r]]) for r in range(len(Kh))), \
        "gang and one-team bits differ"


if "--api" in sys.argv:      # what a caller of the Python mirror sees: BeamSearchCoder(block_size=None).encode / decode of ONE latent, host work included
    from torch.distributions import Normal
    for B in (20, 10):
        q = bench.synthetic_batch(1, eng.device, 0)
        for shared in (True, False):
            c = irec.BeamSearchCoder(kl_per_partition=3., n_beams=B, extra_samples=1.2, block_size=None)
            c.no_split = not shared
            qd, pd = Normal(q[0], q[1], validate_args=False), Normal(q[2], q[3], validate_args=False)
            for _ in range(3):
                idx, sample = c.encode(qd, pd, seed=42)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                idx, sample = c.encode(qd, pd, seed=42)
            torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 10
            out = c.decode(pd, idx, seed=42)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                out = c.decode(pd, idx, seed=42)
            torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 10
            assert torch.equal(out, sample)
            print(f"BeamSearchCoder(block_size=None, n_beams={B}) on one 8192-dim latent, {'gangs' if shared else 'no_split'}: encode {te * 1e3:.3f} ms, "
                  f"decode {td * 1e3:.3f} ms, K = {len(idx)}", flush=True)
    sys.exit(0)
if "--ablate" in sys.argv:
    run("1 x 8192 dims, B=20 (None)", 1, 8192, None, 20, 36, 128, 5)
    run("24 x 8192 dims (one image, None)", 24, 8192, None, 20, 36, 128, 3)
    sys.exit(0)
for st in (1, 2, 3, 4, 6, 9) if "--stripes" in sys.argv else ():
    run("1 x 8192 dims, B=20, stripes<=%d" % st, 1, 8192, None, 20, 36, 128, 5, flags=st << 12)
    run("24 x 8192 dims, B=20, stripes<=%d" % st, 24, 8192, None, 20, 36, 128, 3, flags=st << 12)
    run("1 x 8192 dims, B=10, stripes<=%d" % st, 1, 8192, None, 10, 36, 128, 5, flags=st << 12)
run("1 x 8192 dims, B=20 (None)", 1, 8192, None, 20, 36, 128, 5)
run("1 x 8192 dims, B=10 (None)", 1, 8192, None, 10, 36, 128, 5)
run("1 x 8192 dims, B=30 S=54 (None)", 1, 8192, None, 30, 54, 128, 3)
run("4 x 2048-dim blocks of one latent", 1, 8192, 2048, 20, 36, 64, 5)
run("24 x 8192 dims (one image, None)", 24, 8192, None, 20, 36, 128, 3)
run("64 x 8192 dims (None)", 64, 8192, None, 20, 36, 128, 3)
for nb_ in (96, 128, 192, 256, 320, 384):
    run("%d x 8192 dims (None)" % nb_, nb_, 8192, None, 20, 36, 128, 2)
run("192 x 4096-dim blocks", 96, 8192, 4096, 20, 36, 128, 2)
run("384 x 2048-dim blocks", 96, 8192, 2048, 20, 36, 64, 2)
run("1 x 65536 dims (None)", 1, 65536, None, 20, 36, 1024, 1)
if "--huge" in sys.argv:
    run("1 x 301056 dims (Kodak level 1)", 1, 301056, None, 20, 36, 4096, 1)
