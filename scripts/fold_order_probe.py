"""Probe (GPU box): does a K-aware order of the rows of a mid-size call shorten it?  The team encoder's shared-row hand-out puts
call row w (< n_cu) whole on CU w and the halves of call row n_cu + j on CUs 2j, 2j + 1: the probe lists the rows so that the
costliest ones are the shared ones and meet the cheapest whole rows."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "relative-entropy-coding_amd")]
import irec
from irec.engine import BlockLayout
from oracle import oracle as O
eng = irec.get_engine()


def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for r in range(reps):
        fn(); ev[r + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[r].elapsed_time(ev[r + 1]) for r in range(reps))
    return ts[0], ts[len(ts) // 2]


def sub_layout(lay, rows):
    rows = torch.as_tensor(rows, device=lay.block_dim.device)
    sub = object.__new__(BlockLayout)
    sub.__dict__.update(lay.__dict__)
    sub.block_base, sub.block_pos, sub.block_dim = lay.block_base[rows], lay.block_pos[rows], lay.block_dim[rows]
    sub.natural, sub._natural_dev = None, None
    return sub


for L, B, omega, eps1, shape in ((38, 20, 3.0, 1.2, "default"), (33, 20, 3.0, 1.2, "default"), (45, 20, 3.0, 1.2, "default"),
                                  (50, 20, 3.0, 1.2, "default"), (56, 20, 3.0, 1.2, "default")):
    n, bs = 8192, 1000
    S = int(np.exp(omega * eps1))
    st = [O.synthetic_latent(1234 + i, n) for i in range(L)]
    if os.environ.get("SKEW"):      # per-tensor scale on delta: K differs between the tensors of the call (2 ... ~45)
        rng = np.random.default_rng(77)
        st = [((mp + (mq - mp) * np.float32(np.exp(np.clip(rng.normal(0.0, 0.6), -1.5, 0.9)))).astype(np.float32), sq, mp, sp) for mq, sq, mp, sp in st]
    q = [torch.from_numpy(np.stack([s[k] for s in st])).cuda().contiguous() for k in range(4)]
    lay = eng.layout(L, n, bs, 42)
    params = eng.params(omega, S, B, irec._lib.IREC_FLAG_REUSE_TABLES | irec._lib.IREC_FLAG_SHAPE.get(shape, 0) | irec._lib.IREC_FLAG_LISTED_ORDER)
    by_cost = eng.params(omega, S, B, irec._lib.IREC_FLAG_REUSE_TABLES | irec._lib.IREC_FLAG_SHAPE.get(shape, 0))
    _, K0 = eng.block_kl(params, lay, *q)
    cost = (K0.to(torch.int64).clamp_(min=0) * lay.block_dim.to(torch.int64)).cpu().numpy()
    Kh = K0.cpu().numpy()
    nb, n_cu = lay.n_blocks, 256
    asc = np.argsort(cost, kind="stable")
    e = nb - n_cu
    orders = {"as listed": np.arange(nb)}
    if e > 0:
        orders["fold (cheapest whole, costliest shared, descending)"] = np.concatenate([asc[:n_cu], asc[n_cu:][::-1]])
        orders["costliest whole, cheapest shared"] = np.concatenate([asc[e:], asc[:e]])
        orders["ascending"] = asc
    print(f"== {nb} blocks B={B} S={S} shape={shape}: K histogram {np.bincount(Kh).tolist()} plan {eng.plan(params, lay, 48)}", flush=True)
    ref = None
    orders["the library's cost-ordered hand-out (rows as listed)"] = np.arange(nb)
    for name, rows in orders.items():
        sub = sub_layout(lay, rows)
        pr = by_cost if name.startswith("the library") else params
        mn, med = timed(lambda: eng.encode_blocks(pr, sub, *q, 42, 48))
        K2, idx2, sample = eng.encode_blocks(pr, sub, *q, 42, 48)
        inv = np.argsort(rows)
        Kn, idn = K2.cpu().numpy()[inv], idx2.cpu().numpy()[inv]
        res = (Kn, [idn[r, :Kn[r]].tolist() for r in range(nb)], sample.cpu().numpy())
        same = "-" if ref is None else str(np.array_equal(ref[0], res[0]) and ref[1] == res[1] and np.array_equal(ref[2], res[2]))
        ref = ref or res
        print(f"   {name}: min {mn:.4f} ms, median {med:.4f} ms (tables kept); same outputs as listed order: {same}", flush=True)
    for name, pr in (("listed", eng.params(omega, S, B, irec._lib.IREC_FLAG_SHAPE.get(shape, 0) | irec._lib.IREC_FLAG_LISTED_ORDER)),
                     ("by cost", eng.params(omega, S, B, irec._lib.IREC_FLAG_SHAPE.get(shape, 0)))):
        mn, med = timed(lambda: eng.encode_blocks(pr, lay, *q, 42, 48))
        print(f"   as issued (tables built per call), {name}: min {mn:.4f} ms, median {med:.4f} ms", flush=True)
