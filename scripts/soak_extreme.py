"""Diagnostic soak at the edges of float32: prior scales of 1e-19 .. 1e-15 (variances down to the denormal range) and of
1e15 .. 1e18, posteriors 1e-6 .. 1 times as wide, means at the scale of the prior.  GPU (table and fused encoders, decoder)
against the oracle, bit for bit.  Usage: python scripts/soak_extreme.py [cases]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import irec
from oracle import oracle as O
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(77)
bad = done = skipped = 0
t0 = time.time()
for case in range(n_cases):
    D = int(rng.choice([5, 64, 192, 257, 1000]))
    B = int(rng.choice([1, 5, 10, 20, 50])); omega = float(rng.choice([2.0, 3.0, 4.0])); S = int(np.exp(omega))
    regime = int(rng.integers(0, 4))
    if regime == 0:   lsp = rng.uniform(-19, -15, D)
    elif regime == 1: lsp = rng.uniform(15, 18, D)
    elif regime == 2: lsp = rng.choice([-19.0, -17.0, 0.0, 17.0], D)            # mixed magnitudes in one block
    else:             lsp = rng.uniform(-3, 3, D)
    sp = (10.0 ** lsp)
    mp = sp * rng.normal(0, 1, D)
    ratio = 10.0 ** rng.uniform(-6 if regime == 3 else -2, 0, D)
    sq = sp * ratio
    mq = mp + sp * rng.normal(0, 0.3, D)
    t4 = tuple(a.astype(np.float32) for a in (mq, sq, mp, sp))
    kl = O.block_kl(*t4)
    K = O.num_aux(kl, omega)
    if not np.isfinite(kl) or K > 200 or K * S * B * D > 2e8:
        skipped += 1; continue
    seed = int(rng.integers(0, 2 ** 31))
    ridx, rs = O.encode_block(*t4, seed, omega, S, B, max_K=512)
    q = torch.distributions.Normal(torch.from_numpy(t4[0][None]).cuda(), torch.from_numpy(t4[1][None]).cuda(), validate_args=False)
    p = torch.distributions.Normal(torch.from_numpy(t4[2][None]).cuda(), torch.from_numpy(t4[3][None]).cuda(), validate_args=False)
    for variant in ("table", "fused", "generic"):
        c = irec.BeamSearchCoder(kl_per_partition=omega, n_beams=B, extra_samples=1.0)
        c.team = variant == "table"; c.fused_philox = variant == "fused"; c.force_generic = variant == "generic"
        idx, sample = c.encode(q, p, seed=seed)
        ok = [int(i) for i in idx] == ridx and np.array_equal(sample.cpu().numpy()[0], rs, equal_nan=True)
        ok = ok and torch.equal(c.decode(p, idx, seed=seed), sample)
        if not ok:
            bad += 1
            print(f"MISMATCH case {case} regime {regime} D {D} B {B} S {S} K {K} variant {variant}: gpu {[int(i) for i in idx][:8]} ref {ridx[:8]}", flush=True)
    done += 1
    if case % 50 == 0: print(f"[extreme] case {case}: {done} blocks, {skipped} skipped, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"extreme soak: {done} blocks x 3 variants (+ decode), {skipped} skipped (K out of range), mismatches: {bad}")
