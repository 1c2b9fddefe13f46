#!/usr/bin/env python3
"""How often can K = ceil(KL / Omega) (beam_search_coder.py:57-59) depend on the order / precision of the KL sum?

The reference reduces tfd.kl_divergence with a float32 tf.reduce_sum whose association is TensorFlow's business; the oracle
and the kernels sum float64 terms in a fixed tree and round once.  K differs between two such sums only when KL / Omega
falls within their disagreement of an integer.  For blocks of the bench workload (SURVEY.md §8d statistics, 1000- and
192-dim blocks) this script computes KL five ways -- float64 terms / float64 sum (the oracle's), and float32 terms summed
in float32 sequentially, reversed, pairwise (numpy) and in 8 interleaved lanes -- and reports
  * the distance of KL64 / Omega to the nearest integer (the margin), its smallest values and quantiles,
  * the spread of the float32 sums around KL64 (the noise),
  * the number of blocks whose K differs between any two of the five.

TEST INFRASTRUCTURE (CPU only); merges a "k_margin" object into profiles/margins.json.
Usage: python scripts/k_margins.py [n_latents]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_DIMS, BLOCK = 8192, 1000


def kl_terms(mq, sq, mp, sp, dt):
    """tfd.kl_divergence(Normal, Normal) per dim, TFP 0.9: 0.5 ((mq - mp) / sp)^2 + 0.5 (r - 1 - ln r), r = (sq / sp)^2."""
    mq, sq, mp, sp = (a.astype(dt) for a in (mq, sq, mp, sp))
    d = (mq - mp) / sp
    t = sq / sp
    r = t * t
    return dt(0.5) * d * d + (dt(0.5) * (r - dt(1)) - np.log(t))


def main():
    n_lat = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    rng = np.random.default_rng(1234)
    res = {}
    for omega in (3.0, 5.0):
        margins, noise, flips, n_blocks = [], [], 0, 0
        for i0 in range(0, n_lat, 256):
            n = min(256, n_lat - i0)
            mp = rng.normal(0, 1, (n, N_DIMS)); lsp = rng.normal(0, 0.25, (n, N_DIMS)); sp = np.exp(lsp)
            mq = mp + sp * rng.normal(0, 0.2, (n, N_DIMS)); sq = np.exp(lsp - np.abs(rng.normal(0, 0.05, (n, N_DIMS))))
            q = [a.astype(np.float32) for a in (mq, sq, mp, sp)]
            t64 = kl_terms(*q, np.float64)
            t32 = kl_terms(*q, np.float32)
            for lo in range(0, N_DIMS, BLOCK):
                hi = min(lo + BLOCK, N_DIMS)
                a64, a32 = t64[:, lo:hi], t32[:, lo:hi]
                kl64 = a64.sum(axis=1)
                sums = [np.float32(kl64).astype(np.float64)]                               # float64 sum rounded once (oracle)
                sums.append(np.cumsum(a32, axis=1, dtype=np.float32)[:, -1].astype(np.float64))           # sequential
                sums.append(np.cumsum(a32[:, ::-1], axis=1, dtype=np.float32)[:, -1].astype(np.float64))  # reversed
                sums.append(a32.sum(axis=1, dtype=np.float32).astype(np.float64))                          # pairwise
                pad = (-a32.shape[1]) % 8
                lanes = np.pad(a32, ((0, 0), (0, pad))).reshape(n, -1, 8)
                sums.append(np.cumsum(lanes, axis=1, dtype=np.float32)[:, -1, :].sum(axis=1, dtype=np.float32).astype(np.float64))
                ks = np.stack([np.ceil(np.float32(s) / np.float32(omega)) for s in sums])
                flips += int((ks.max(axis=0) != ks.min(axis=0)).sum())
                x = kl64 / omega
                margins.append(np.abs(x - np.round(x)))
                noise.append(np.max(np.abs(np.stack(sums[1:]) - kl64), axis=0) / omega)
                n_blocks += n
        m = np.concatenate(margins); nz = np.concatenate(noise)
        res[f"omega_{omega:g}"] = {
            "blocks": n_blocks,
            "blocks_with_K_depending_on_the_sum": flips,
            "margin_quantiles": {str(qq): float(np.quantile(m, qq)) for qq in (0.0, 1e-4, 1e-3, 1e-2, 0.5)},
            "noise_over_omega_quantiles": {str(qq): float(np.quantile(nz, qq)) for qq in (0.5, 0.99, 1.0)},
            "expected_flip_fraction": float(2 * np.quantile(nz, 0.99)),   # margin is uniform on [0, 0.5]
        }
        print(omega, json.dumps(res[f"omega_{omega:g}"]))
    path = os.path.join(ROOT, "profiles", "margins.json")
    d = json.load(open(path))
    d["k_margin"] = {"what": __doc__.split("\n\n")[1].replace("\n", " "), "n_latents": n_lat, **res}
    json.dump(d, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
