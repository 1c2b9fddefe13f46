"""Diagnostic: rate of encode_generic_kernel (blocks of more than 1024 dims, 60 < B <= 64) next to the team encoder's on the
same latents -- what a caller pays for block_size > 1024.  Usage: [LATENTS=512] [SKIP_GENERIC=1] python scripts/generic_rate.py
Prints look-ups per clock per CU next to the rate (S * D * (1 + (K - 1) * B) look-ups per block).  One block of 8192 dims
holds a team for ~33 ms: the rate of a call is the rate of its teams only when every team slot has several blocks (LATENTS=3072)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import bench, irec
eng = irec.get_engine()
L = int(os.environ.get('LATENTS', '512'))
q = bench.synthetic_batch(L, eng.device, 0)
# (round 4: block sizes above 1024 take encode_chunk_kernel; FORCE_GENERIC pins the generic kernel for the comparison)
for bs, B, flags in ((1000, 20, 0), (2048, 20, 0), (2048, 20, 1), (4096, 20, 0), (4096, 20, 1), (None, 20, 0), (None, 20, 1), (2048, 10, 0),
                     (None, 10, 0), (1000, 64, 0)):
    if flags == 1 and os.environ.get("SKIP_GENERIC"):
        continue
    lay = eng.layout(L, bench.N_DIMS, bs, bench.SEED)
    params = eng.params(3.0, 36, B, flags, table_steps=128)
    plan = eng.plan(params, lay, 128)
    eng.encode_blocks(params, lay, *q, bench.SEED, 128)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        K, idx, s = eng.encode_blocks(params, lay, *q, bench.SEED, 128)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    Kh = K.cpu().numpy().astype(np.int64); dims = lay.block_dim.cpu().numpy().astype(np.int64)
    evals = float((36 * dims * (1 + np.maximum(Kh - 1, 0) * B) * (Kh > 0)).sum())
    print(f"block_size {str(bs):>5s} B {B:2d}: {plan['kernel']:28s} {dt * 1e3:9.2f} ms for {L} latents -> {L / dt:9.0f} latents/s, "
          f"{evals / dt / (plan['n_cu'] * plan['clock_mhz'] * 1e6):5.2f} look-ups/clk/CU, max K {int(Kh.max())}", flush=True)
