#!/usr/bin/env python3
"""Cross-check of the three CPU restatements of BeamSearchCoder.encode_block and the top-B margin histogram
(SURVEY.md §7 "hard parts": rank-B vs rank-B+1 score gap against float32 summation noise).

Three implementations that share NO arithmetic code for the score:
  canonical : oracle/irec_oracle.c, quadratic form  C_b + sum_d (G_bd + H_d z) z  in the fixed fma tree (what the GPU runs)
  literal   : oracle/irec_oracle.c, TFP's log_prob difference op by op, sequential float32 sum
  torch     : oracle/ref_shaped_torch.py, torch.special.ndtri / torch reductions / torch.argsort on [S,B,1,D] tensors
For every block and step it records
  * the gap between the B-th and (B+1)-th best score (canonical), absolute and relative to the score range of the step,
  * the largest difference between the literal and canonical scores of the step over its 2B best candidates, after removing
    the per-step constant the canonical form drops (|(lit_f - lit_best) - (can_f - can_best)|),
and whether the three implementations emit the same indices.  A flip between implementations (or against TensorFlow's own
reduction order) needs gap < noise; the histogram says how often that can happen.

TEST INFRASTRUCTURE (reads oracle/); writes profiles/margins.json.  Usage: python scripts/margins.py [n_blocks_per_setting]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O            # noqa: E402
from oracle import ref_shaped_torch as R  # noqa: E402

SETTINGS = [(3.0, 1.2, 20), (3.0, 1.0, 10), (5.0, 1.0, 30), (6.0, 1.0, 10)]   # (Omega, 1+eps, B): SURVEY.md §7 step 1


def random_block(rng, D, regime):
    mp = rng.normal(0, 1, D)
    lsp = rng.normal(0, 0.25, D)
    sp = np.exp(lsp)
    if regime == 0:      # SURVEY §8d statistics
        mq = mp + sp * rng.normal(0, 0.2, D); sq = np.exp(lsp - np.abs(rng.normal(0, 0.05, D)))
    elif regime == 1:    # sharper posteriors: more partitions
        mq = mp + sp * rng.normal(0, 0.5, D); sq = np.exp(lsp - np.abs(rng.normal(0, 0.3, D)))
    else:                # nearly uninformative dims mixed with a few informative ones
        mq = mp + sp * rng.normal(0, 0.05, D) * (rng.random(D) < 0.9) + sp * rng.normal(0, 1.0, D) * (rng.random(D) < 0.1)
        sq = np.exp(lsp - np.abs(rng.normal(0, 0.02, D)))
    return tuple(a.astype(np.float32) for a in (mq, sq, mp, sp))


def step_margins(trace, S, B):
    """per step: (gap_B, score range, Bcur)"""
    out = []
    Bcur = 1
    for t in range(trace["K"]):
        N = S * Bcur
        sc = np.sort(trace["score"][t][:N].astype(np.float64))[::-1]
        Bnew = min(B, N)
        gap = float(sc[Bnew - 1] - sc[Bnew]) if N > Bnew else float("inf")
        out.append((gap, float(sc[0] - sc[-1]), Bcur))
        Bcur = Bnew
    return out


def main():
    n_per = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rng = np.random.default_rng(20261003)
    rec = {"settings": [], "n_blocks": 0, "n_steps": 0, "index_mismatch_literal": 0, "index_mismatch_torch": 0}
    gaps, rel_gaps, noise, regimes = [], [], [], []
    t_start = time.time()
    for omega, eps1, B in SETTINGS:
        S = O.n_samples(omega, eps1)
        n_here = n_per if S < 200 else max(n_per // 4, 8)     # the S = 403 settings are 10x the work per step
        st = {"omega": omega, "eps1": eps1, "B": B, "S": S, "blocks": 0, "steps": 0, "mismatch_literal": 0,
              "mismatch_torch": 0, "min_gap": None, "max_noise": 0.0, "steps_gap_below_noise": 0}
        for k in range(n_here):
            D = int(rng.choice([1000, 1000, 192, int(rng.integers(1, 1025))]))
            mq, sq, mp, sp = random_block(rng, D, k % 3)
            seed = int(rng.integers(0, 2 ** 31 - 1))
            ci, cs, ctr = O.encode_block(mq, sq, mp, sp, seed, omega, S, B, O.CANONICAL, trace=True)
            li, ls, ltr = O.encode_block(mq, sq, mp, sp, seed, omega, S, B, O.LITERAL, trace=True)
            ti, ts = R.encode_block(mq, sq, mp, sp, seed, omega, S, B)
            st["blocks"] += 1
            st["mismatch_literal"] += int(ci != li)
            st["mismatch_torch"] += int(ci != ti)
            if ci != li:
                continue   # after a flip the two runs follow different beams: their later steps are not comparable
            m = step_margins(ctr, S, B)
            Bcur = 1
            for t, (gap, rng_t, _) in enumerate(m):
                N = S * Bcur
                can = ctr["score"][t][:N].astype(np.float64)
                lit = ltr["score"][t][:N].astype(np.float64)
                # noise where it matters: over the 2B best candidates (hopeless candidates have scores of -1e4 and worse,
                # whose float32 sums are noisy but can never reach the top B)
                top = np.argsort(-can)[:min(N, 2 * B)]
                nz = float(np.max(np.abs((lit[top] - lit[top[0]]) - (can[top] - can[top[0]]))))
                if np.isfinite(gap):
                    gaps.append(gap); rel_gaps.append(gap / max(rng_t, 1e-30)); noise.append(nz); regimes.append(k % 3)
                    st["steps"] += 1
                    st["min_gap"] = gap if st["min_gap"] is None else min(st["min_gap"], gap)
                    st["max_noise"] = max(st["max_noise"], nz)
                    st["steps_gap_below_noise"] += int(gap <= 2 * nz)
                Bcur = min(B, N)
        rec["settings"].append(st)
        rec["n_blocks"] += st["blocks"]; rec["n_steps"] += st["steps"]
        rec["index_mismatch_literal"] += st["mismatch_literal"]; rec["index_mismatch_torch"] += st["mismatch_torch"]
        print(f"[margins] Omega={omega} eps1={eps1} B={B} S={S}: {st}  ({time.time() - t_start:.0f} s)", flush=True)
    gaps, noise = np.array(gaps), np.array(noise)
    edges = [0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0, 10.0, np.inf]
    rec["gap_histogram"] = {"edges": [str(e) for e in edges], "counts": np.histogram(gaps, bins=edges)[0].tolist()}
    rec["noise_histogram"] = {"edges": [str(e) for e in edges], "counts": np.histogram(noise, bins=edges)[0].tolist()}
    rec["gap_quantiles"] = {str(q): float(np.quantile(gaps, q)) for q in (0.0, 0.001, 0.01, 0.1, 0.5)}
    rec["noise_quantiles"] = {str(q): float(np.quantile(noise, q)) for q in (0.5, 0.9, 0.99, 1.0)}
    rec["gap_over_noise_quantiles"] = {str(q): float(np.quantile(gaps / np.maximum(noise, 1e-30), q)) for q in (0.0, 0.001, 0.01, 0.1, 0.5)}
    regimes = np.array(regimes)
    names = {0: "SURVEY 8d statistics (the bench workload)", 1: "sharp posteriors, K up to ~100 (scores of magnitude 1e2..1e3: "
             "the literal float32 sum cancels catastrophically)", 2: "mostly uninformative dims"}
    rec["by_regime"] = {names[r]: {"steps": int((regimes == r).sum()),
                                   "gap_quantiles": {str(q): float(np.quantile(gaps[regimes == r], q)) for q in (0.0, 0.01, 0.5)},
                                   "noise_quantiles": {str(q): float(np.quantile(noise[regimes == r], q)) for q in (0.5, 0.99, 1.0)},
                                   "steps_gap_below_2x_noise": int((gaps[regimes == r] <= 2 * noise[regimes == r]).sum())}
                        for r in (0, 1, 2) if (regimes == r).any()}
    rec["note"] = ("gap = score[rank B] - score[rank B+1] per step (canonical mode); noise = max over the 2B best candidates of the "
                   "literal-vs-canonical score difference after removing the per-step constant.  A step can flip between "
                   "float32 summation orders only if gap <~ noise.")
    with open(os.path.join(ROOT, "profiles", "margins.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: rec[k] for k in ("n_blocks", "n_steps", "index_mismatch_literal", "index_mismatch_torch",
                                          "gap_quantiles", "noise_quantiles", "gap_over_noise_quantiles")}, indent=1))


if __name__ == "__main__":
    main()
