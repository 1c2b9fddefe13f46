"""Diagnostic (round 5): the chunked encoder on 3072 / 6144 one-block latents of 8192 dims per call, as listed and longest first --
how much of a call is its tail (a block holds its team for 40 ms).  Usage: python scripts/chunk_tail.py"""
import os, sys, time
import numpy as np, torch
ROOT = "/root/repo" if os.path.isdir("/root/repo/relative-entropy-coding_amd") else os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import bench, irec
eng = irec.get_engine()
for L in (3072, 6144):
    q = bench.synthetic_batch(L, eng.device, 0)
    for bs, B in ((None, 20),):
        lay = eng.layout(L, bench.N_DIMS, bs, bench.SEED)
        params = eng.params(3.0, 36, B, 0, table_steps=128)
        plan = eng.plan(params, lay, 128)
        for kw in ({}, {"order_by_K": True}):
            eng.encode_blocks(params, lay, *q, bench.SEED, 128, **kw)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(2):
                K, idx, s = eng.encode_blocks(params, lay, *q, bench.SEED, 128, **kw)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
            Kh = K.cpu().numpy().astype(np.int64); dims = lay.block_dim.cpu().numpy().astype(np.int64)
            evals = float((36 * dims * (1 + np.maximum(Kh - 1, 0) * B) * (Kh > 0)).sum())
            print(f"L {L} block_size {bs} B {B} {kw}: {plan['kernel']} {dt*1e3:.2f} ms -> {L/dt:.0f} latents/s, {evals/dt/(plan['n_cu']*plan['clock_mhz']*1e6):.2f} look-ups/clk/CU, K {Kh.min()}..{Kh.max()}", flush=True)
    del q
