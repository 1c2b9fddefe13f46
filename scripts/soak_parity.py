"""Randomised parity soak (GPU box): random block shapes / coder settings / statistics, every encoder variant against
the CPU oracle.  Not part of the pytest suite (minutes of oracle time); prints a summary and exits non-zero on mismatch."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import irec
from oracle import oracle as O

n_cases = int(os.environ.get("SOAK_CASES", "150"))
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
eng = irec.get_engine()
bad, done, t0 = [], 0, time.time()
stats = {"K": [], "evals": 0}
for case in range(n_cases):
    D = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 192, 255, 256, 257, 511, 777, 1000, 1023, 1024, int(rng.integers(1, 1025))]))
    B = int(rng.choice([1, 2, 7, 10, 11, 20, 21, 30, 32, 33, 50, 60, 61, 64, int(rng.integers(1, 65))]))   # (round 3: up to IREC_MAX_BEAMS)
    omega = float(rng.choice([1.0, 2.0, 3.0, 4.0, float(rng.uniform(0.7, 5.0))]))
    eps1 = float(rng.choice([1.0, 1.2, 1.5]))
    if os.environ.get("SOAK_BIG"):   # bias towards more than 1024 candidates per step (streamed top-B, sample passes, keys in the slab)
        B = int(rng.choice([1, 10, 11, 16, 20, 21, 30, 31, 32, 40, 50, 60])); omega = float(rng.choice([4.0, 5.0, 5.5, 6.0])); eps1 = float(rng.choice([1.0, 1.1, 1.2]))
        D = int(rng.choice([192, 777, 1000, 1024]))
    if os.environ.get("SOAK_LARGE"):   # round 4: blocks of more than 1024 dims (chunked encoder for B <= 20, generic beyond) and up to 256 beams
        D = int(rng.choice([1025, 1279, 1500, 2048, 2049, 2500, 3000, 4096, 6000, 9217, int(rng.integers(1025, 4200)), int(rng.integers(1025, 12000))]))
        B = int(rng.choice([1, 2, 7, 10, 11, 20, 20, 10, 21, 25, 30, 31, 32, 33, 41, 50, 60, 61, 100, 256])); omega = float(rng.choice([2.0, 3.0, 3.5, float(rng.uniform(1.0, 3.6))]))
        eps1 = float(rng.choice([1.0, 1.2]))
    if os.environ.get("SOAK_TEN"):   # round 6: the ten-beam encoder (irec_ten.hip): 2 <= B <= 10, S * 10 <= 256, every block size up to 1024, extreme statistics
        B = int(rng.integers(2, 11)); omega = float(rng.choice([0.5, 1.0, 2.0, 3.0, 3.2, float(rng.uniform(0.1, 3.2))])); eps1 = 1.0
        D = int(rng.choice([1, 2, 3, 4, 5, 63, 64, 65, 192, 255, 256, 257, 511, 512, 513, 767, 768, 769, 777, 1000, 1023, 1024, int(rng.integers(1, 1025))]))
    S = int(np.exp(omega * eps1))
    if S * B * D > (1.3e7 if (os.environ.get("SOAK_BIG") or os.environ.get("SOAK_LARGE")) else 6e6):          # keep the oracle fast
        continue
    NT = int(rng.choice([1, 2, 3, 5, 8] + ([16, 40, 261, 300, 384] if os.environ.get("SOAK_TEN") else [])))   # tensors per call (= blocks per call: both teams of a CU get work)
    def draw():
        style = rng.integers(0, 5)
        mp = rng.normal(0, 1, D); lsp = rng.normal(0, 0.5, D); sp = np.exp(lsp)
        if style == 0:   # benign
            mq = mp + sp * rng.normal(0, 0.2, D); sq = np.exp(lsp - np.abs(rng.normal(0, 0.05, D)))
        elif style == 1: # tight posteriors -> many partitions
            mq = mp + sp * rng.normal(0, 1.0, D); sq = sp * rng.uniform(0.05, 0.5, D)
        elif style == 2: # posterior wider than prior on some dims
            mq = mp + sp * rng.normal(0, 0.3, D); sq = sp * rng.uniform(0.5, 1.5, D)
        elif style == 3: # extreme scales
            sp = np.exp(rng.normal(0, 3.0, D)); mq = mp + sp * rng.normal(0, 0.5, D); sq = sp * rng.uniform(0.2, 1.0, D)
        else:            # nearly identical q and p (K small, near ties)
            mq = mp + sp * rng.normal(0, 0.02, D); sq = sp * np.exp(rng.normal(0, 0.01, D))
        return tuple(a.astype(np.float32) for a in (mq, sq, mp, sp)), style
    tens, Ks, style = [], [], 0
    for _ in range(NT):
        for _try in range(20):
            t4, style = draw()
            K = O.num_aux(O.block_kl(*t4), omega)
            if K <= 300 and K * S * B * D <= (1.5e9 if os.environ.get("SOAK_LARGE") else 4e8):
                break
        else:
            continue
        tens.append(t4); Ks.append(K)
    if not tens:
        continue
    seed = int(rng.integers(0, 2 ** 31))
    refs = [O.encode_block(*t4, seed, omega, S, B, max_K=512, margins=True) for t4 in tens]
    mq, sq, mp, sp = (np.stack([t4[k] for t4 in tens]) for k in range(4))
    q = torch.distributions.Normal(torch.from_numpy(mq).cuda(), torch.from_numpy(sq).cuda(), validate_args=False)
    p = torch.distributions.Normal(torch.from_numpy(mp).cuda(), torch.from_numpy(sp).cuda(), validate_args=False)
    # (SOAK_LARGE, round 5: calls this small are coded by gangs of teams -- "alone" pins one team per block)
    for variant in (("table", "alone", "generic") if os.environ.get("SOAK_LARGE") else ("table", "one_table", "fused", "generic")):
        c = irec.BeamSearchCoder(kl_per_partition=omega, n_beams=B, extra_samples=eps1)
        c.n_samples = S
        c.force_generic = variant == "generic"; c.fused_philox = variant == "fused"; c.one_table = variant == "one_table"; c.team = variant in ("table", "alone")
        c.no_split = variant == "alone"
        idx, sample = c.encode(q, p, seed=seed, batched=True)
        sh = sample.cpu().numpy()
        ok = all([int(i) for i in idx[n]] == refs[n][0] and np.array_equal(sh[n], refs[n][1]) for n in range(len(tens)))
        if ok and min(Ks):
            ok = torch.equal(c.decode(p, idx, seed=seed, batched=True), sample)
        if not ok:
            bad.append((case, variant, D, B, S, omega, max(Ks), style, seed))
    if os.environ.get("SOAK_MARGINS"):   # round 5: irec_beam_encode_ex -- same outputs, and the four margin floats of every block against the oracle's
        lay = eng.layout(len(tens), D, None, seed)
        qd = tuple(torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (mq, sq, mp, sp))
        mk = max(max(Ks), 1)
        Kd, idd, smp, mg = eng.encode_blocks_margins(eng.params(omega, S, B), lay, *qd, seed, mk)
        Kh, ih, mh, sh2 = Kd.cpu().numpy(), idd.cpu().numpy(), mg.cpu().numpy(), smp.cpu().numpy()
        for n in range(len(tens)):
            r = lay.natural[n]
            if not (ih[r, :Kh[r]].tolist() == refs[n][0] and np.array_equal(sh2[n], refs[n][1]) and np.array_equal(mh[r], refs[n][2])):
                bad.append((case, "margins", D, B, S, omega, max(Ks), style, seed))
                break
    K = max(Ks)
    for Kn in Ks:
        stats["evals"] += S * D * (1 + max(Kn - 1, 0) * B)
    done += len(tens); stats["K"].append(K)
    if case % 100 == 99:   # a long run must keep writing: the GPU box takes minutes of silence for a hang
        print(f"[soak] case {case + 1}/{n_cases}: {done} blocks, {len(bad)} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"soak: {done} random blocks x {3 if os.environ.get('SOAK_LARGE') else 4} variants in {time.time() - t0:.0f} s; K range {min(stats['K'])}..{max(stats['K'])}; "
      f"{stats['evals'] / 1e9:.2f} G proposal evals checked; mismatches: {len(bad)}")
for b in bad[:20]:
    print("MISMATCH case=%d variant=%s D=%d B=%d S=%d omega=%.3f K=%d style=%d seed=%d" % b)
sys.exit(1 if bad else 0)
