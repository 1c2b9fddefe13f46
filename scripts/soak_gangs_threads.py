"""Stress (GPU box): gang calls of random shapes issued back to back from several threads on their own streams -- the calls contend for the CUs
(all members of a gang must be resident), so barriers, give-ups (out_K = -2) and the poison path are exercised under load.  Every call's
result is compared, on the device, with the same call on one team per block (IREC_FLAG_NO_SPLIT); a block reported as not coded (-2) is
counted, not compared.  SOAK_MODE=small: the same for the cooperative encoders of blocks of at most 1024 dims (split encoder, shared rows).
Usage: [SOAK_THREADS=3] [SOAK_CALLS=120] [SOAK_SEED=1] [SOAK_MODE=small] python scripts/soak_gangs_threads.py"""
import os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import irec
from irec import _lib
from oracle import oracle as O

eng = irec.get_engine()
small = os.environ.get("SOAK_MODE", "") == "small"
n_threads, n_calls, seed0 = int(os.environ.get("SOAK_THREADS", "3")), int(os.environ.get("SOAK_CALLS", "120")), int(os.environ.get("SOAK_SEED", "1"))
stats = {"calls": 0, "blocks": 0, "gave_up_calls": 0, "mismatch": [], "errors": [], "gang_calls": 0}
lock = threading.Lock()
pool = [O.synthetic_latent(4000 + i, 20000) for i in range(8)]       # statistics to cut shapes from


def work(w):
    rng = np.random.default_rng(seed0 * 100 + w)
    try:
        with torch.cuda.stream(torch.cuda.Stream()):
            for call in range(n_calls):
                if small:   # SOAK_MODE=small: blocks of at most 1024 dims -- the split encoder (< 64 blocks) and the shared rows of the team encoder
                    n = int(rng.choice([8192, 8192, 3000, 12288])); bs = int(rng.choice([1000, 1000, 1024, 500]))
                    n_t = int(rng.choice([1, 1, 2, 3, 5, 7, 14, 20, 28, 38, 42]))
                    B = int(rng.choice([7, 10, 10, 16, 20, 20])); S = int(rng.choice([7, 20, 36]))
                else:
                    n = int(rng.choice([1025, 2048, 3000, 4097, 5000, 8192, 12000, int(rng.integers(1025, 20000))]))
                    n_t = int(rng.choice([1, 1, 2, 3, 5, 8, 16, 30]))
                    if n * n_t > 200000:
                        n_t = max(1, 200000 // n)
                    B = int(rng.choice([1, 7, 10, 20, 20, 30, 32, 50]))
                    bs = None if rng.random() < 0.6 else int(rng.choice([1500, 2048, 4096]))
                    S = int(rng.choice([7, 20, 36]))
                seed = int(rng.integers(0, 2 ** 31))
                off = int(rng.integers(0, 20000 - n + 1))
                q = tuple(torch.from_numpy(np.stack([pool[(w + i) % 8][k][off:off + n] for i in range(n_t)])).cuda().contiguous() for k in range(4))
                lay = eng.layout(n_t, n, bs, seed)
                if (lay.max_dim <= 1024) != small:
                    continue
                gang, alone = eng.params(3.0, S, B), eng.params(3.0, S, B, _lib.IREC_FLAG_NO_SPLIT)
                is_gang = eng.plan(gang, lay, 256)["split"] >= 2
                K, idx, smp = eng.encode_blocks(gang, lay, *q, seed, 256)
                K1, idx1, smp1 = eng.encode_blocks(alone, lay, *q, seed, 256)
                Kh, K1h = K.cpu().numpy(), K1.cpu().numpy()
                gave_up = bool((Kh == -2).any())
                ok = True
                if not gave_up:
                    ih, i1 = idx.cpu().numpy(), idx1.cpu().numpy()
                    ok = np.array_equal(Kh, K1h) and torch.equal(smp, smp1) and all(np.array_equal(ih[r, :Kh[r]], i1[r, :Kh[r]]) for r in range(len(Kh)))
                with lock:
                    stats["calls"] += 1; stats["blocks"] += lay.n_blocks; stats["gave_up_calls"] += int(gave_up); stats["gang_calls"] += int(is_gang)
                    if not ok:
                        stats["mismatch"].append((w, call, n, n_t, B, bs, S, seed))
                    if stats["calls"] % 50 == 0:
                        print(f"[soak gangs] {stats['calls']} calls, {stats['blocks']} blocks, {stats['gang_calls']} shared, {stats['gave_up_calls']} gave up, "
                              f"{len(stats['mismatch'])} mismatches", flush=True)
    except Exception as e:                      # noqa: BLE001
        with lock:
            stats["errors"].append((w, repr(e)))


t0 = time.time()
th = [threading.Thread(target=work, args=(w,)) for w in range(n_threads)]
for t in th:
    t.start()
for t in th:
    t.join()
print(f"soak gangs: {n_threads} threads, {stats['calls']} calls ({stats['gang_calls']} by gangs), {stats['blocks']} blocks in {time.time() - t0:.0f} s; "
      f"gave up: {stats['gave_up_calls']} calls; mismatches: {len(stats['mismatch'])}; errors: {len(stats['errors'])}")
for m in stats["mismatch"][:10]:
    print("MISMATCH thread=%d call=%d n=%d n_t=%d B=%d bs=%s S=%d seed=%d" % m)
for e in stats["errors"][:5]:
    print("ERROR", e)
sys.exit(1 if stats["mismatch"] or stats["errors"] else 0)
