#!/usr/bin/env python3
"""How often does another float32 summation order move an emitted index?  Measured, at scale, next to what the margins predict.

The encoder's indices depend on the scores only through the top-B set of every step but the last and the winner of the last step
(beam_search_coder.py:85-89,118-122).  TensorFlow's reduce_sum order is not reproducible here, so this script
runs the two CPU restatements that share no score arithmetic -- CANONICAL (the fixed fma tree the GPU runs) and LITERAL (TFP's
log_prob difference op by op, sequential float32 sum: a summation order as far from the tree as any) -- over >= 10^5 blocks of the
bench workload (SURVEY.md §8d statistics, RVAE latents of 8192 dims in blocks of 1000, B = 20, Omega = 3, S = 36) and records
  * blocks whose index lists differ (a flip), blocks whose K differs,
  * every block's margins (irec_oracle_encode_block_ex: what irec_beam_encode_ex reports from the device),
  * the flip rate inside each margin bucket, and the exposure the margins predict: the fraction of blocks whose smallest
    comparison (min of the set gap and the winner's lead) lies below the summation noise measured by scripts/margins.py,
  * optionally the same against oracle/ref_shaped_torch.py (torch reductions, torch.argsort) on the first --torch-latents latents.

TEST INFRASTRUCTURE (CPU only; reads oracle/).  Writes profiles/margins_flips.json.
Usage: python scripts/margins_flips.py [--latents 11200] [--threads 8] [--torch-latents 300]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O            # noqa: E402

OMEGA, EPS1, B, SEED, N, BS = 3.0, 1.2, 20, 42, 8192, 1000
EDGES = [0.0, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 1e-2, 1e-1, 1.0, np.inf]
NOISE_P99, NOISE_MAX = 4.6e-5, 6.5e-5     # literal-vs-canonical score noise of the bench regime, profiles/margins.json (by_regime)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latents", type=int, default=11200)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--torch-latents", type=int, default=300)
    ap.add_argument("--chunk", type=int, default=224)
    ap.add_argument("--first-image", type=int, default=100000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "margins_flips.json"))
    ap.add_argument("--beams", type=int, default=B, help="n_beams (default: the headline's 20)")
    ap.add_argument("--omega", type=float, default=OMEGA)
    ap.add_argument("--eps1", type=float, default=EPS1, help="1 + extra_samples exponent (default 1.2: S = 36 at Omega = 3)")
    args = ap.parse_args()
    globals().update(B=args.beams, OMEGA=args.omega, EPS1=args.eps1)
    S = O.n_samples(OMEGA, EPS1)
    gaps, tops, flips, kdiff, lit_small = [], [], [], [], []
    t0 = time.time()
    done = 0
    while done < args.latents:
        n = min(args.chunk, args.latents - done)
        st = [O.synthetic_latent(args.first_image + done + i, N) for i in range(n)]
        a = [np.stack([s[k] for s in st]) for k in range(4)]
        ci, cs, _, cm = O.encode_tensors_omp(*a, SEED, OMEGA, S, B, BS, mode=O.CANONICAL, n_threads=args.threads, margins=True)
        li, ls, _, lm = O.encode_tensors_omp(*a, SEED, OMEGA, S, B, BS, mode=O.LITERAL, n_threads=args.threads, margins=True)
        for i in range(n):
            for j in range(len(ci[i])):
                gaps.append(cm[i, j, 0]); tops.append(cm[i, j, 2])
                same_K = len(ci[i][j]) == len(li[i][j])
                kdiff.append(not same_K)
                flips.append(same_K and ci[i][j] != li[i][j])
                lit_small.append(min(lm[i, j, 0], lm[i, j, 2]))
        done += n
        el = time.time() - t0
        print(f"[margins_flips] {done} latents = {len(gaps)} blocks, {int(np.sum(flips))} flips, {int(np.sum(kdiff))} K differences, "
              f"{el:.0f} s ({len(gaps) / el:.0f} blocks/s both modes)", flush=True)
    gaps, tops, flips, kdiff = np.array(gaps, np.float64), np.array(tops, np.float64), np.array(flips), np.array(kdiff)
    small = np.minimum(gaps, tops)                                  # the closest comparison that decides an index of the block
    rec = {"what": __doc__.split("\n\n")[1].replace("\n", " "),
           "settings": {"omega": OMEGA, "extra_samples": EPS1, "n_beams": B, "n_samples": S, "dims": N, "block_size": BS, "seed": SEED,
                        "statistics": "SURVEY.md 8d (oracle.synthetic_latent), images %d..%d" % (args.first_image, args.first_image + done - 1)},
           "blocks": int(len(gaps)), "latents": int(done),
           "index_flips_literal_vs_canonical": int(flips.sum()), "K_differs_literal_vs_canonical": int(kdiff.sum()),
           "measured_flip_rate_per_block": float(flips.mean()),
           "set_gap_quantiles": {str(q): float(np.quantile(gaps[np.isfinite(gaps)], q)) for q in (0.0, 1e-4, 1e-3, 1e-2, 0.1, 0.5)},
           "winner_lead_quantiles": {str(q): float(np.quantile(tops[np.isfinite(tops)], q)) for q in (0.0, 1e-4, 1e-3, 1e-2, 0.1, 0.5)},
           "exact_ties_at_a_deciding_comparison": int((small == 0).sum()),
           "noise_thresholds": {"p99": NOISE_P99, "max": NOISE_MAX, "source": "profiles/margins.json by_regime (bench statistics)"},
           "predicted_exposure": {"blocks_below_noise_p99": float((small < NOISE_P99).mean()), "blocks_below_noise_max": float((small < NOISE_MAX).mean()),
                                  "blocks_below_2x_noise_max": float((small < 2 * NOISE_MAX).mean())},
           "buckets": []}
    for lo, hi in zip(EDGES[:-1], EDGES[1:]):
        m = (small >= lo) & (small < hi) if lo > 0 else (small >= 0) & (small < hi)
        rec["buckets"].append({"closest_comparison": [lo, hi if np.isfinite(hi) else "inf"], "blocks": int(m.sum()), "flips": int(flips[m].sum()),
                               "flip_rate": float(flips[m].mean()) if m.any() else None})
    fl = small[flips]
    rec["closest_comparison_of_the_flipped_blocks"] = {"max": float(fl.max()) if fl.size else None, "median": float(np.median(fl)) if fl.size else None,
                                                        "n": int(fl.size)}
    rec["seconds"] = time.time() - t0
    if args.torch_latents > 0:
        from oracle import ref_shaped_torch as R
        import torch
        torch.set_num_threads(max(1, args.threads or os.cpu_count() or 1))
        tf, tb, t1 = 0, 0, time.time()
        for i in range(min(args.torch_latents, done)):
            st = O.synthetic_latent(args.first_image + i, N)
            ti, _ = R.encode_tensor(*st, SEED, OMEGA, S, B, BS)
            ci, _ = O.encode_tensor(*st, SEED, OMEGA, S, B, block_size=BS)
            tb += len(ci)
            tf += sum(1 for a_, b_ in zip(ti, ci) if list(a_) != list(b_))
            if (i + 1) % 50 == 0:
                print(f"[margins_flips] torch: {i + 1} latents, {tf} of {tb} blocks differ, {time.time() - t1:.0f} s", flush=True)
        rec["ref_shaped_torch_vs_canonical"] = {"blocks": tb, "index_flips": tf, "measured_flip_rate_per_block": tf / max(tb, 1)}
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: rec[k] for k in rec if k not in ("what", "buckets")}, indent=1))
    for b in rec["buckets"]:
        print(b)


if __name__ == "__main__":
    main()
