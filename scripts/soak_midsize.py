"""Randomised parity soak of the mid-size call policies (GPU box): calls of 64 ... ~520 blocks with random beams, samples, table window
and per-tensor K skew -- the default policy (rows shared between teams, rows dealt by cost, whichever applies) against the same call with
IREC_FLAG_NO_SPLIT | IREC_FLAG_LISTED_ORDER (every row on one team, as listed) bit for bit, and against the CPU oracle on two tensors per
case.  Not part of the pytest suite; exits non-zero on a mismatch."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import irec
from oracle import oracle as O
F = irec._lib
eng = irec.get_engine()
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "5")))
n_cases = int(os.environ.get("SOAK_CASES", "60"))
bad, policies, t0 = [], {}, time.time()
for case in range(n_cases):
    n = int(rng.choice([8192, 8192, 4096, 3000, 2048]))
    bs = int(rng.choice([1000, 1000, 512, 777]))
    bpt = -(-n // bs)
    n_lat = int(rng.integers(max(2, 64 // bpt + 1), 520 // bpt + 1))
    B = int(rng.choice([7, 10, 11, 16, 20, 20]))
    omega = float(rng.choice([2.0, 3.0, 3.0, 4.0]))
    eps1 = float(rng.choice([1.0, 1.2]))
    S = int(np.exp(omega * eps1))
    steps = int(rng.choice([0, 0, 6, 12]))
    skew = float(rng.choice([0.0, 0.3, 0.7]))
    stats = []
    for i in range(n_lat):
        mq, sq, mp, sp = O.synthetic_latent(int(rng.integers(0, 1 << 30)), n)
        f = np.float32(np.exp(np.clip(rng.normal(0.0, skew), -1.5, 0.8))) if skew else np.float32(1.0)
        stats.append(((mp + (mq - mp) * f).astype(np.float32), sq, mp, sp))
    if rng.random() < 0.3:
        k0 = int(rng.integers(0, n_lat)); stats[k0] = (stats[k0][2].copy(), stats[k0][3].copy(), stats[k0][2], stats[k0][3])   # KL = 0
    q = [torch.from_numpy(np.stack([s[k] for s in stats])).cuda().contiguous() for k in range(4)]
    seed = int(rng.integers(0, 1 << 31))
    lay = eng.layout(n_lat, n, bs, seed)
    max_K = 96
    dflt = eng.params(omega, S, B, 0, table_steps=steps)
    plain = eng.params(omega, S, B, F.IREC_FLAG_NO_SPLIT | F.IREC_FLAG_LISTED_ORDER, table_steps=steps)
    plan = eng.plan(dflt, lay, max_K)
    key = (plan["kernel"], plan["split"])
    policies[key] = policies.get(key, 0) + 1
    K, idx, sample = eng.encode_blocks(dflt, lay, *q, seed, max_K)
    K2, idx2, sample2 = eng.encode_blocks(plain, lay, *q, seed, max_K)
    Kh, K2h = K.cpu().numpy(), K2.cpu().numpy()
    ok = np.array_equal(Kh, K2h) and Kh.min() >= 0 and torch.equal(sample, sample2)
    ih, ih2 = idx.cpu().numpy(), idx2.cpu().numpy()
    coded = Kh <= max_K
    ok = ok and all(np.array_equal(ih[r, :Kh[r]], ih2[r, :Kh[r]]) for r in range(lay.n_blocks) if coded[r])
    if ok and coded.all():
        ok = torch.equal(eng.decode_blocks(dflt, lay, q[2], q[3], seed, K, idx), sample)
    for i in rng.choice(n_lat, size=min(2, n_lat), replace=False) if ok else []:
        rows = [lay.natural[int(i) * lay.blocks_per_tensor + j] for j in range(lay.blocks_per_tensor)]
        if not all(coded[r] for r in rows) or max(int(Kh[r]) for r in rows) > 40:
            continue
        ridx, rs = O.encode_tensor(*stats[int(i)], seed, omega, S, B, block_size=bs)
        ok = ok and [ih[r, :Kh[r]].tolist() for r in rows] == ridx and np.array_equal(sample[int(i)].cpu().numpy(), rs)
    if not ok:
        bad.append((case, n, bs, n_lat, B, S, steps, skew, seed, key))
    if case % 10 == 9:
        print(f"[soak-midsize] {case + 1}/{n_cases}: {len(bad)} mismatches, {time.time() - t0:.0f} s", flush=True)
print("policies met:", {f"{k[0]} W={k[1]}": v for k, v in sorted(policies.items())})
print(f"soak-midsize: {n_cases} calls in {time.time() - t0:.0f} s; mismatches: {len(bad)}")
for b in bad[:20]:
    print("MISMATCH", b)
sys.exit(1 if bad else 0)
