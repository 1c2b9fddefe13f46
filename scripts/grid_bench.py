"""Throughput of every cell of the reference's own hyper-parameter sweep (examples/lossless/data_aggregation.py:5-7:
kl_per_partition 2..6 x extra_samples {1, 1.1, 1.2, 1.5} x n_beams {1, 10, 50}) on RVAE-shape latents (8192 dims, blocks of
1000): latents/s, the block kernel irec_encode_plan names, and look-ups per clock per CU (E = S * D * (1 + (K - 1) * B) proposal
evaluations per block).  One oracle-checked latent per cell.  Usage: python scripts/grid_bench.py [--latents 256]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import irec  # noqa: E402


GATHER_PEAK = 13.70   # look-ups/clk/CU: random addresses, 2-choice bank assignment (profiles/r01j/gather_rates.log)


def arithmetic_floor(kernel, S, B, D=1000):
    """Look-ups per clock per CU that the ARITHMETIC of one block-step allows (DESIGN.md §4 "The sweep grid"), a ceiling in
    the unit of the measured column: live look-ups S * B * D of a steady-state step over the larger of
      * the gather pipe:  S * B' * D / 13.70 cycles, B' = beam SLOTS the build scores (phantoms included),
      * VALU issue of the busiest SIMD (it hosts one wave per beam stripe of the team, plus wave 0's selection): 2 cycles per
        wave instruction; per wave and step  480 (IEEE step constants of four dims: six divisions, a square root) + 45 per beam
        of the update + 60 (combine, barriers), wave 0 another 650 (selection); per wave and sample 8 per beam (address add,
        packed fma; the look-up itself issues on the LDS port) + 45 (20-value reduce-scatter, row handling);
      * for the one-wave-per-block encoder: 1 900 cycles of step constants + 75 S of the gather pipe per block-step.
    Latency is NOT in it: what a cell loses to its dependent phases with few teams in flight is the gap to this number."""
    if kernel.startswith("encode_lone"):
        return S * D / (1900.0 + 75.0 * S)
    nb, teams, bs = (int(x) for x in kernel.split("<")[1].split(">")[0].split(",")[:3])
    per_stripe = nb // bs
    live_stripes = max(1, min(bs, -(-B // per_stripe)))                       # stripes holding at least one live beam
    slots = sum(per_stripe if 2 * min(per_stripe, max(0, B - k * per_stripe)) >= per_stripe else min(per_stripe, max(0, B - k * per_stripe))
                for k in range(live_stripes)) if B > 1 else 1
    lds = S * slots * D / GATHER_PEAK
    valu = 2.0 * (live_stripes * (480 + 60) + 45 * min(B, slots) + 650 + (8 * slots + 45 * live_stripes) * S)
    return S * B * D / max(lds, valu)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latents", type=int, default=256)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--beams", type=str, default="1,10,50")
    ap.add_argument("--omegas", type=str, default="2,3,4,5,6")
    ap.add_argument("--eps", type=str, default="1.0,1.1,1.2,1.5")
    ap.add_argument("--check", type=int, default=1, help="latents per cell compared with the oracle (0: none)")
    ap.add_argument("--budget-s", type=float, default=900.0)
    ap.add_argument("--shape", type=str, default="default", help="IREC_FLAG_SHAPE_* name (diagnostics; 'team' pins the team encoder for one-beam calls)")
    a = ap.parse_args()
    from oracle import oracle as O
    eng = irec.get_engine()
    n, bs = 8192, 1000
    stats = [O.synthetic_latent(1000 + i, n) for i in range(a.latents)]
    ql, qs, pl, ps = (torch.from_numpy(np.stack([s[k] for s in stats])).cuda().contiguous() for k in range(4))
    lay = eng.layout(a.latents, n, bs, 42)
    dims = lay.block_dim.cpu().numpy().astype(np.int64)
    t_start = time.time()
    print(f"# latents per call {a.latents} ({lay.n_blocks} blocks); columns: Omega 1+eps B S | kernel | ms/call | latents/s | "
          f"G look-ups/s | look-ups/clk/CU | mean K | oracle check | ms/call with the tables kept | look-ups/clk/CU then | "
          f"arithmetic floor (look-ups/clk/CU) | tables-kept rate / floor", flush=True)
    worst = None
    for omega in [float(x) for x in a.omegas.split(",")]:
        for eps1 in [float(x) for x in a.eps.split(",")]:
            S = int(np.exp(omega * eps1))
            for B in [int(x) for x in a.beams.split(",")]:
                if time.time() - t_start > a.budget_s:
                    print("# time budget reached", flush=True)
                    return
                params = eng.params(omega, S, B, irec._lib.IREC_FLAG_SHAPE[a.shape])
                max_K = 24
                K, idx, sample = eng.encode_blocks(params, lay, ql, qs, pl, ps, 42, max_K)   # warm-up (tables, scratch)
                torch.cuda.synchronize()
                Kh = K.cpu().numpy().astype(np.int64)
                assert Kh.max() <= max_K and Kh.min() >= 0, (Kh.min(), Kh.max())
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.reps + 1)]
                ev[0].record()
                for r in range(a.reps):
                    eng.encode_blocks(params, lay, ql, qs, pl, ps, 42, max_K)
                    ev[r + 1].record()
                torch.cuda.synchronize()
                ms = min(ev[r].elapsed_time(ev[r + 1]) for r in range(a.reps))
                # the same call with the proposal tables kept across calls (IREC_FLAG_REUSE_TABLES: what the Python coder does
                # by default -- the tables depend on (seed, S, D) only): the block kernel alone
                keep = eng.params(omega, S, B, irec._lib.IREC_FLAG_REUSE_TABLES | irec._lib.IREC_FLAG_SHAPE[a.shape])
                eng.encode_blocks(keep, lay, ql, qs, pl, ps, 42, max_K)
                ev[0].record()
                for r in range(a.reps):
                    eng.encode_blocks(keep, lay, ql, qs, pl, ps, 42, max_K)
                    ev[r + 1].record()
                torch.cuda.synchronize()
                ms_keep = min(ev[r].elapsed_time(ev[r + 1]) for r in range(a.reps))
                plan = eng.plan(params, lay, max_K)
                E = float((S * dims * (1 + np.maximum(Kh - 1, 0) * B) * (Kh > 0)).sum())
                lps = E / (ms * 1e-3)
                per_clk = lps / (plan["n_cu"] * plan["clock_mhz"] * 1e6)
                per_clk_keep = E / (ms_keep * 1e-3) / (plan["n_cu"] * plan["clock_mhz"] * 1e6)
                ok = "-"
                if a.check:
                    ih = idx.cpu().numpy()
                    ok = "ok"
                    for i in range(a.check):
                        ridx, rs = O.encode_tensor(*stats[i], 42, omega, S, B, block_size=bs)
                        got = [ih[lay.natural[i * 9 + j], :Kh[lay.natural[i * 9 + j]]].tolist() for j in range(9)]
                        if got != ridx or not np.array_equal(sample[i].cpu().numpy(), rs):
                            ok = "MISMATCH"
                line = (f"{omega:g} {eps1:g} {B:2d} {S:5d} | {plan['kernel']:38s} | {ms:9.3f} | {a.latents / ms * 1e3:10.1f} | "
                        f"{lps / 1e9:8.1f} | {per_clk:6.2f} | {Kh.mean():5.2f} | {ok} | {ms_keep:9.3f} | {per_clk_keep:6.2f} | "
                        f"{arithmetic_floor(plan['kernel'], S, B):6.2f} | {per_clk_keep / arithmetic_floor(plan['kernel'], S, B):5.2f}")
                print(line, flush=True)
                if worst is None or per_clk < worst[0]:
                    worst = (per_clk, line)
    print("# worst cell:", worst[1] if worst else None, flush=True)


if __name__ == "__main__":
    main()
