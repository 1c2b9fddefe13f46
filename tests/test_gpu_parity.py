"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the committed
golden fixtures.  Bar: bit-exact indices, K and codelength; bit-exact samples (tolerance stated by BASELINE.json's
north_star is 1e-5 on reconstructions -- we require equality and also assert the 1e-5 bound explicitly)."""
import os
import time

import ctypes

import numpy as np
import pytest
import torch

from conftest import golden_files

pytestmark = pytest.mark.gpu


def _normal(loc, scale, device="cuda"):
    return torch.distributions.Normal(torch.as_tensor(loc, device=device), torch.as_tensor(scale, device=device),
                                      validate_args=False)


# teams per CU over three table copies (forced also for small calls) / the default choice between that and the one-table
# encoder by call size / one-table proposal-table kernel / Philox-fused kernel / fallback
VARIANTS = ["table", "auto", "one_table", "one_table_nosplit", "fused", "generic"]


def _coder(omega, B, eps1, block_size=None, variant="table"):
    import irec
    c = irec.BeamSearchCoder(kl_per_partition=omega, n_beams=B, extra_samples=eps1, block_size=block_size)
    c.force_generic = variant == "generic"
    c.fused_philox = variant == "fused"
    c.one_table = variant in ("one_table", "one_table_nosplit")
    c.no_split = variant == "one_table_nosplit"     # small calls otherwise take the split encoder (W workgroups per block)
    c.team = variant == "table"
    return c


def test_native_library_is_loaded(engine):
    import irec
    maps = open("/proc/self/maps").read()
    assert "libirec_hip.so" in maps
    assert engine.ctx and b"gfx950" in irec._lib.load().irec_version()


@pytest.mark.parametrize("width", [64, 32])
def test_reduce_scatter_is_the_canonical_tree(engine, width):
    rng = np.random.default_rng(width)
    x = (rng.standard_normal((64, width)) * np.exp(rng.uniform(-3, 3, (64, width)))).astype(np.float32)
    got = engine.test_reduce_scatter(torch.from_numpy(x).cuda()).cpu().numpy()
    for lane in range(64):
        col = lane * width // 64
        part = x[:, col].copy()
        step = 32
        while step >= 1:
            part[:step] = part[:step] + part[step:2 * step]
            step >>= 1
        assert got[lane] == part[0], (lane, got[lane], part[0])


@pytest.mark.parametrize("width", [20, 10, 21])
def test_arbitrary_width_reduce_scatter_is_the_canonical_tree(engine, width):
    """The team encoder's 20-/10-value reduce-scatter: every column's total comes out of the same tree (lanes paired at
    distance 32, 16, 8, 4, 2, 1); the trailing all-reduce stages leave each column's total in 2 (4) neighbouring lanes."""
    scoring_form = width == 21      # reduce_scatter_20: register pairs + bank-masked DPP adds, other value-to-lane map
    width = 20 if scoring_form else width
    rng = np.random.default_rng(100 + width)
    x = (rng.standard_normal((64, width)) * np.exp(rng.uniform(-3, 3, (64, width)))).astype(np.float32)
    out = engine.test_reduce_scatter(torch.from_numpy(x).cuda(), scoring_form=scoring_form).cpu().numpy()
    tot, owner = out[:64], out[64:].astype(np.int64)
    for col in range(width):
        part = x[:, col].copy()
        step = 32
        while step >= 1:
            part[:step] = part[:step] + part[step:2 * step]
            step >>= 1
        lanes = np.nonzero(owner == col)[0]
        # 20 -> 10 -> 5 -> 3 -> 2 -> 1 leaves one all-reduce stage (2 owners); 10 -> 5 -> 3 -> 2 -> 1 -> 1 two (4 owners)
        n_own = 2 if width == 20 else 4
        assert len(lanes) == n_own and (lanes >> (n_own // 2)).min() == (lanes >> (n_own // 2)).max(), (col, lanes)
        assert (tot[lanes] == part[0]).all(), (col, tot[lanes], part[0])
    assert ((owner >= -1) & (owner < width)).all()


def _ref_select(scores, n_select, bcur):
    # tf.argsort(DESCENDING) == top_k: value descending, ties -> lower index; NaN after every number
    keyed = sorted(range(len(scores)), key=lambda i: (np.isnan(scores[i]), -scores[i] if not np.isnan(scores[i]) else 0.0, i))
    return [(i // bcur, i % bcur) for i in keyed[:n_select]]


@pytest.mark.parametrize("case", ["random720", "tiny", "ties_small", "tie_storm", "nan_inf_zero", "n1024", "n4440_stream",
                                  "n2960_stream", "n4096_stream", "n1025_stream", "n3000_ties", "n3000_storm", "n2000_front",
                                  "n2000_nan"])
def test_top_b_selection(engine, case):
    rng = np.random.default_rng(5)
    if case == "random720":
        sc, B, bcur = rng.standard_normal(720).astype(np.float32) * 30, 20, 20
    elif case == "tiny":
        sc, B, bcur = rng.standard_normal(36).astype(np.float32), 20, 1
    elif case == "ties_small":
        sc, B, bcur = rng.integers(-3, 3, 720).astype(np.float32), 20, 20        # many exact ties, resolved by index
    elif case == "tie_storm":
        sc, B, bcur = np.full(720, -7.25, np.float32), 20, 20                      # > 64 survivors -> fallback scan
    elif case == "nan_inf_zero":
        sc = rng.standard_normal(300).astype(np.float32)
        sc[::7] = np.nan; sc[3] = np.inf; sc[11] = -np.inf; sc[20] = 0.0; sc[21] = -0.0
        B, bcur = 30, 10
    elif case == "n1024":
        sc, B, bcur = rng.standard_normal(1024).astype(np.float32), 64, 32
    # N > 1024: wave 0 streams the keys (16-byte reads + scalar tail) instead of holding them in registers
    elif case == "n2960_stream":
        sc, B, bcur = rng.standard_normal(2960).astype(np.float32) * 9, 20, 20
    elif case == "n4096_stream":
        sc, B, bcur = rng.standard_normal(4096).astype(np.float32), 32, 32
    elif case == "n1025_stream":
        sc, B, bcur = rng.standard_normal(1025).astype(np.float32), 30, 5
    elif case == "n3000_ties":
        sc, B, bcur = rng.integers(-40, 40, 3000).astype(np.float32), 30, 30     # ties across the wave boundaries
    elif case == "n3000_storm":
        sc, B, bcur = np.full(3000, 1.5, np.float32), 30, 30                      # > 64 survivors -> scan
    elif case == "n2000_front":
        sc = rng.standard_normal(2000).astype(np.float32); sc[:40] += 100.0       # all winners in the first rows
        B, bcur = 30, 20
    elif case == "n2000_nan":
        sc = rng.standard_normal(2000).astype(np.float32); sc[::3] = np.nan; sc[1500:] = -np.inf
        B, bcur = 30, 20
    else:
        sc, B, bcur = rng.standard_normal(4440).astype(np.float32), 30, 30
    got = engine.test_select(torch.from_numpy(sc).cuda(), B, bcur).cpu().numpy().tolist()
    assert [tuple(g) for g in got] == _ref_select(sc.tolist(), B, bcur)
    got = engine.test_select(torch.from_numpy(sc).cuda(), B, bcur, quick=True).cpu().numpy().tolist()   # (round 4: the quick form)
    assert [tuple(g) for g in got] == _ref_select(sc.tolist(), B, bcur)
    if len(sc) > 1024:   # the streamed selection reads 16 bytes at a time only from a 16-byte-aligned key array
        got = engine.test_select(torch.from_numpy(sc).cuda(), B, bcur, key_offset=1).cpu().numpy().tolist()
        assert [tuple(g) for g in got] == _ref_select(sc.tolist(), B, bcur)


def test_in_kernel_philox_stream(engine, oracle):
    for seed, n in [(42, 36 * 1000), (43, 4097), (0, 64), (2 ** 31 - 1, 64), (69420 + 3, 403)]:
        got = engine.device_uniform_int(seed, n).cpu().numpy()
        assert np.array_equal(got, oracle.uniform_int(seed, n)), seed


@pytest.mark.parametrize("dim,S,steps", [(1000, 36, 3), (192, 36, 2), (1, 5, 2), (130, 20, 2), (1024, 7, 1), (999, 11, 2)])
def test_proposal_table_dlogs_and_bank_spread(engine, oracle, dim, S, steps):
    """tab[t][s][d] = dlog_g(r) + 10006 c: the dlog part must be the discrete log of the reference's int32 draw
    (beam_search_coder.py:38-43), c in {0, 1}, and c must spread every 32-lane look-up group over the LDS banks."""
    P = 10007
    g = next(c for c in range(2, P) if len({pow(c, e, P) for e in range(P - 1)}) == P - 1)   # smallest primitive root
    dlog = np.zeros(P, dtype=np.int64)
    x = 1
    for e in range(P - 1):
        dlog[x] = e
        x = x * g % P
    tab = engine.test_proposal_table(42, S, dim, steps).cpu().numpy().view(np.uint16).astype(np.int64)
    dp = tab.shape[2]
    busiest_plain, busiest = [], []
    for t in range(steps):
        r = oracle.uniform_int(42 + t, S * dim).reshape(S, dim)
        want = dlog[r]
        got = tab[t, :, :dim]
        assert np.array_equal(got % (P - 1), want), t
        assert ((got // (P - 1)) <= 1).all()
        assert (tab[t, :, dim:] % (P - 1) == 0).all()   # padding dims: entry 0 of either copy
        for s in range(S):
            row = np.concatenate([tab[t, s], np.zeros(-dp % 128, dtype=np.int64)])[: (dp + 127) // 128 * 128]
            valid = np.arange(row.size) < dp
            for m in range(row.size // 128):          # 32 lanes x 4 dim slots
                for i in range(4):
                    sel = slice(128 * m + i, 128 * m + 128, 4)
                    v = row[sel][valid[sel]]
                    if v.size:
                        busiest.append(np.bincount(v % 32, minlength=32).max())
                        busiest_plain.append(np.bincount((v % (P - 1)) % 32, minlength=32).max())
    # exact optimum of the two-choice assignment: never worse than leaving every look-up in copy 0
    assert np.mean(busiest) <= np.mean(busiest_plain)
    if dim >= 999:
        assert np.mean(busiest) < 2.3 and np.mean(busiest_plain) > 3.2, (np.mean(busiest), np.mean(busiest_plain))


def test_block_kl_and_partition_count(engine, oracle):
    n_t, n, bs = 4, 8192, 1000
    stats = [oracle.synthetic_latent(i, n) for i in range(n_t)]
    ql, qs, pl, ps = (torch.from_numpy(np.stack([s[k] for s in stats])).cuda().contiguous() for k in range(4))
    lay = engine.layout(n_t, n, bs, 42)
    params = engine.params(3.0, 36, 20)
    kl, K = engine.block_kl(params, lay, ql, qs, pl, ps)
    kl, K = kl.cpu().numpy(), K.cpu().numpy()
    perm = oracle.tf_shuffle_perm(42, n)
    for t in range(n_t):
        for b, (lo, hi) in enumerate(oracle.split_blocks(n, bs)):
            g = perm[lo:hi]
            ref = np.float32(oracle.block_kl(*(s[g] for s in stats[t])))
            row = lay.natural[t * lay.blocks_per_tensor + b]
            assert kl[row] == ref, (t, b, kl[row], ref)
            assert K[row] == oracle.num_aux(ref, 3.0)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("path", golden_files("block"), ids=os.path.basename)
def test_golden_blocks(engine, path, variant):
    g = np.load(path)
    c = _coder(float(g["kl_per_partition"]), int(g["n_beams"]), float(g["extra_samples"]), variant=variant)
    assert c.n_samples == int(g["n_samples"])
    q = _normal(g["q_loc"][None], g["q_scale"][None])
    p = _normal(g["p_loc"][None], g["p_scale"][None])
    idx, sample = c.encode(q, p, seed=int(g["seed"]))
    assert [int(i) for i in idx] == g["indices"].tolist()
    assert len(idx) == int(g["K"])
    got = sample.cpu().numpy()[0]
    assert np.abs(got - g["sample"]).max(initial=0.0) <= 1e-5
    assert np.array_equal(got, g["sample"])
    assert abs(c.get_codelength(idx) - float(g["codelength"])) < 1e-9
    rec = c.decode(p, idx, seed=int(g["seed"]))
    assert torch.equal(rec, sample)


@pytest.mark.parametrize("variant", VARIANTS)
def test_golden_tensor_rvae_shape(engine, variant):
    g = np.load(golden_files("tensor")[0])
    c = _coder(float(g["kl_per_partition"]), int(g["n_beams"]), float(g["extra_samples"]),
               block_size=int(g["block_size"]), variant=variant)
    q = _normal(g["q_loc"], g["q_scale"])
    p = _normal(g["p_loc"], g["p_scale"])
    idx, sample = c.encode(q, p, seed=int(g["seed"]))
    assert [len(i) for i in idx] == g["K"].tolist()
    for r, i in enumerate(idx):
        assert i == g["indices"][r, :len(i)].tolist(), r
    assert sample.shape == q.loc.shape
    assert np.array_equal(sample.cpu().numpy(), g["sample"])
    rec = c.decode(p, idx, seed=int(g["seed"]))
    assert torch.equal(rec, sample)
    total = sum(c.get_codelength(i) for i in idx)
    assert abs(total - float(g["codelength"])) < 1e-9


def test_reference_unit_test_case(engine):
    # rec/coding/tests/test_coder.py:12-21, written the way the reference writes it
    import irec
    encoder = irec.BeamSearchCoder(kl_per_partition=6., n_beams=10, extra_samples=1.)
    t = torch.distributions.Normal(loc=torch.tensor([[5.1]]), scale=torch.tensor([[0.001]]))
    p = torch.distributions.Normal(loc=torch.tensor([[0.]]), scale=torch.tensor([[1.]]))
    indices, sample = encoder.encode(t, p, seed=69420, update_sampler=False)
    reconstructed_sample = encoder.decode(p, indices, seed=69420)
    assert torch.allclose(sample, reconstructed_sample) and torch.equal(sample, reconstructed_sample)
    assert sample.device.type == "cpu" and len(indices) == 4


def test_cpu_tensors_and_cuda_tensors_agree(engine, oracle):
    mq, sq, mp, sp = oracle.synthetic_latent(11, 500)
    c = _coder(3.0, 20, 1.2)
    i1, s1 = c.encode(_normal(mq[None], sq[None], "cpu"), _normal(mp[None], sp[None], "cpu"), seed=7)
    i2, s2 = c.encode(_normal(mq[None], sq[None]), _normal(mp[None], sp[None]), seed=7)
    assert i1 == i2 and torch.equal(s1, s2.cpu()) and s1.device.type == "cpu" and s2.device.type == "cuda"


@pytest.mark.parametrize("variant", VARIANTS)
def test_batched_latents_match_oracle(engine, oracle, variant):
    n_t, shape, bs = 3, (16, 16, 32), 1000
    stats = [oracle.synthetic_latent(50 + i, 8192) for i in range(n_t)]
    ql, qs, pl, ps = (np.stack([s[k].reshape(shape) for s in stats]) for k in range(4))
    c = _coder(3.0, 20, 1.2, block_size=bs, variant=variant)
    idx, sample = c.encode(_normal(ql, qs), _normal(pl, ps), seed=42, batched=True)
    assert len(idx) == n_t and sample.shape == (n_t,) + shape
    for t in range(n_t):
        ridx, rs = oracle.encode_tensor(ql[t], qs[t], pl[t], ps[t], 42, 3.0, 36, 20, block_size=bs)
        assert idx[t] == ridx
        assert np.array_equal(sample[t].cpu().numpy(), rs)
    rec = c.decode(_normal(pl, ps), idx, seed=42, batched=True)
    assert torch.equal(rec, sample)


def test_large_block_uses_generic_path(engine, oracle):
    # block_size=None on a 3000-dim tensor: D > 1024 -> the chunked encoder (round 4), and the generic kernel when pinned
    mq, sq, mp, sp = oracle.synthetic_latent(77, 3000)
    ridx, rs = oracle.encode_block(mq, sq, mp, sp, 3, 3.0, 20, 10)
    for generic in (False, True):
        c = _coder(3.0, 10, 1.0)
        c.force_generic = generic
        idx, sample = c.encode(_normal(mq[None], sq[None]), _normal(mp[None], sp[None]), seed=3)
        assert [int(i) for i in idx] == ridx and np.array_equal(sample.cpu().numpy()[0], rs), generic
        assert torch.equal(c.decode(_normal(mp[None], sp[None]), idx, seed=3), sample)


@pytest.mark.parametrize("n,bs,omega,eps1,B,n_t", [(8192, 2048, 3.0, 1.2, 20, 6), (8192, 4096, 3.0, 1.2, 20, 4), (8192, None, 3.0, 1.2, 20, 5),
                                                   (8192, None, 3.0, 1.0, 10, 3), (5000, 2048, 3.0, 1.0, 7, 3), (3001, None, 2.0, 1.5, 1, 4),
                                                   (1025, None, 3.0, 1.2, 20, 3), (2500, 2048, 3.0, 1.2, 13, 3), (12288, None, 3.0, 1.0, 10, 2),
                                                   (4099, 1100, 3.5, 1.0, 20, 2),
                                                   # round 5: up to 32 beams (beam passes of 10 / 16 on one team), blocks beyond 16 384 dims
                                                   (8192, None, 3.0, 1.2, 30, 2), (8192, 2048, 3.0, 1.2, 32, 3), (3000, None, 3.0, 1.0, 25, 2),
                                                   (2600, 1300, 3.0, 1.2, 21, 5), (17000, None, 3.0, 1.0, 10, 1),
                                                   # ... and up to 60 beam slots (passes of 10, two teams): 32 < B <= 60 of blocks beyond 1024 dims
                                                   (3000, None, 2.0, 1.0, 50, 2), (2500, 2048, 3.0, 1.0, 40, 3), (2500, None, 3.0, 1.2, 60, 2)])
def test_blocks_beyond_1024_dims_take_the_chunked_encoder(engine, oracle, n, bs, omega, eps1, B, n_t):
    """Coder.__init__ takes any block_size, None included (coder.py:29-36,415-419: the whole tensor as ONE block -- the
    reference's default).  Round 4: such blocks are walked in chunks of 1024 dims by encode_chunk_kernel over the team
    encoder's tables instead of falling to the generic kernel.  Indices, K and samples against the oracle, bit for bit:
    block_size 2048 / 4096 / None on 8192-dim tensors, ragged chunks and dim groups (5000, 3001, 1025, 4099 dims), a call that
    mixes blocks above and below 1024 dims (2500 = 2048 + 452), one beam, beam counts that are not a build's, a 12 288-dim
    block; decode(encode) exact; the generic kernel pinned gives the same bits; irec_encode_plan names the kernel.
    Round 5: the steady-state scoring is the team encoder's software pipeline; beams are scored in passes of 10 (16 for 32 slots) whose
    partials are combined pass by pass, so THREE teams per CU fit next to the table copies (12 waves at 168 VGPRs); 20 < B <= 32;
    blocks of up to 65 536 dims (a 17 000-dim block: K = 123)."""
    S = oracle.n_samples(omega, eps1)
    stats = [oracle.synthetic_latent(8100 + i, n) for i in range(n_t)]
    ql, qs, pl, ps = (torch.from_numpy(np.stack([s[k] for s in stats])).cuda().contiguous() for k in range(4))
    lay = engine.layout(n_t, n, bs, 42)
    max_K = 160
    params = engine.params(omega, S, B, table_steps=max_K)      # (tables over every partition: nothing is left to the second pass)
    plan = engine.plan(params, lay, max_K)
    want = ("encode_chunk_kernel<10,10,3>" if B <= 10 else "encode_chunk_kernel<20,10,3>" if B <= 20 else
            "encode_chunk_kernel<30,10,3>" if B <= 30 else "encode_chunk_kernel<32,16,2>" if B <= 32 else
            "encode_chunk_kernel<%d,10,2>" % (-(-B // 10) * 10))
    import irec
    if plan["split"] >= 2:    # (calls this small are coded by gangs of teams: the three-team build, else the one-team build of the beam count)
        assert plan["kernel"] == (want[:-1] + ",gang>" if want.endswith(",10,3>") else want[:-2] + "1,gang>"), plan["kernel"]
    else:
        assert plan["kernel"] == want, plan["kernel"]
    assert plan["table_kernel"] == "prep_kernel (copy bits)" and plan["lds_bytes"] <= 160 * 1024
    K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, max_K)
    Kh, ih = K.cpu().numpy(), idx.cpu().numpy()
    assert Kh.min() >= 0 and Kh.max() <= max_K, (int(Kh.min()), int(Kh.max()))
    if plan["split"] >= 2:                                      # ... and every block on ONE team gives the same bits
        alone = engine.params(omega, S, B, irec._lib.IREC_FLAG_NO_SPLIT, table_steps=max_K)
        assert engine.plan(alone, lay, max_K)["kernel"] == want
        K1, idx1, sample1 = engine.encode_blocks(alone, lay, ql, qs, pl, ps, 42, max_K)
        assert torch.equal(K, K1) and torch.equal(sample, sample1) and all(np.array_equal(ih[r, :Kh[r]], idx1.cpu().numpy()[r, :Kh[r]]) for r in range(lay.n_blocks))
    gen = engine.params(omega, S, B, irec._lib.IREC_FLAG_FORCE_GENERIC)
    K2, idx2, sample2 = engine.encode_blocks(gen, lay, ql, qs, pl, ps, 42, max_K)
    assert torch.equal(K, K2) and torch.equal(sample, sample2)
    ih2 = idx2.cpu().numpy()
    for r in range(lay.n_blocks):
        assert np.array_equal(ih[r, :Kh[r]], ih2[r, :Kh[r]]), r
    assert torch.equal(engine.decode_blocks(params, lay, pl, ps, 42, K, idx), sample)
    bpt = lay.blocks_per_tensor
    for i in sorted({0, n_t - 1}):
        ridx, rs = oracle.encode_tensor(*stats[i], 42, omega, S, B, block_size=bs)
        got = [ih[lay.natural[i * bpt + j], :Kh[lay.natural[i * bpt + j]]].tolist() for j in range(bpt)]
        if bs is None:
            assert got[0] == ridx, i
        else:
            assert got == ridx, i
        assert np.array_equal(sample[i].cpu().numpy(), rs), i


def test_chunked_encoder_second_pass_and_batches(engine, oracle):
    """Steps beyond the proposal-table window: the chunked encoder draws their rows in the kernel (round 5: Philox + discrete log per
    quad and sample, copy bit 0 -- the generic kernel's second pass took such blocks until then); K = 0 blocks return p.loc; a batch
    of more blocks than resident teams pulls from the counters."""
    n, bs, S, B = 4096, 2048, 36, 20
    n_t = 300                                              # 600 blocks > 512 resident teams
    stats = [oracle.synthetic_latent(8200 + i, n) for i in range(n_t)]
    ql, qs, pl, ps = (torch.from_numpy(np.stack([s[k] for s in stats])).cuda().contiguous() for k in range(4))
    ql[5] = pl[5]; qs[5] = ps[5]                           # posterior == prior: KL = 0, K = 0
    lay = engine.layout(n_t, n, bs, 42)
    full = engine.params(3.0, S, B)
    short = engine.params(3.0, S, B, table_steps=8)        # K ~ 15 per 2048-dim block: every block codes its last steps from the draw itself
    K, idx, sample = engine.encode_blocks(full, lay, ql, qs, pl, ps, 42, 64)
    K2, idx2, sample2 = engine.encode_blocks(short, lay, ql, qs, pl, ps, 42, 64)
    Kh = K.cpu().numpy()
    assert Kh.max() > 8 and torch.equal(K, K2) and torch.equal(sample, sample2)
    ih, ih2 = idx.cpu().numpy(), idx2.cpu().numpy()
    for r in range(lay.n_blocks):
        assert np.array_equal(ih[r, :Kh[r]], ih2[r, :Kh[r]]), r
    bpt = lay.blocks_per_tensor
    assert all(Kh[lay.natural[5 * bpt + j]] == 0 for j in range(bpt)) and torch.equal(sample[5], pl[5])
    assert torch.equal(engine.decode_blocks(full, lay, pl, ps, 42, K, idx), sample)
    for i in (0, 5, 150, n_t - 1):
        mq, sq, mp, sp = (t[i].cpu().numpy() for t in (ql, qs, pl, ps))
        if i == 5:
            continue
        ridx, rs = oracle.encode_tensor(mq, sq, mp, sp, 42, 3.0, S, B, block_size=bs)
        assert [ih[lay.natural[i * bpt + j], :Kh[lay.natural[i * bpt + j]]].tolist() for j in range(bpt)] == ridx, i
        assert np.array_equal(sample[i].cpu().numpy(), rs), i


def test_blocks_of_any_size_stay_off_the_generic_kernel(engine, oracle):
    """Round 5: `Coder.__init__(block_size=None)` on a tensor of any size (coder.py:29-36,415-419) -- a block of 70 000 dims (beyond round
    4's 16 384 and this round's first 65 536), posteriors close to the prior so that the oracle finishes: the chunked encoder, its scratch
    slabs capped by IREC_SLAB_BYTES_MAX, a table window far shorter than K so that most steps draw their rows in the kernel; and the same
    for 30 beams (passes of 10)."""
    import irec
    rng = np.random.default_rng(77)
    for n, B, eps1, delta in ((70000, 20, 1.2, 0.02), (40000, 30, 1.0, 0.035), (301056, 20, 1.2, 0.01)):   # (the last: Kodak level 1 as ONE block)
        S = oracle.n_samples(3.0, eps1)
        mp = rng.normal(0, 1, n).astype(np.float32); sp = np.exp(rng.normal(0, 0.25, n)).astype(np.float32)
        mq = (mp + sp * rng.normal(0, delta, n)).astype(np.float32); sq = (sp * np.exp(-np.abs(rng.normal(0, 0.005, n)))).astype(np.float32)
        ridx, rs = oracle.encode_block(mq, sq, mp, sp, 11, 3.0, S, B)
        assert 4 <= len(ridx) <= 12, len(ridx)
        lay = engine.layout(1, n, None, 11)
        params = engine.params(3.0, S, B, table_steps=2)
        plan = engine.plan(params, lay, 32)
        assert plan["kernel"].startswith("encode_chunk_kernel<%d," % (20 if B == 20 else 30)) and plan["table_steps"] == 2, plan
        assert plan["workspace_bytes"] < 20 * 2 ** 30
        q = tuple(torch.from_numpy(a[None]).cuda().contiguous() for a in (mq, sq, mp, sp))
        K, idx, sample = engine.encode_blocks(params, lay, *q, 11, 32)
        assert int(K.cpu()[0]) == len(ridx) and idx.cpu().numpy()[0, :len(ridx)].tolist() == ridx
        assert np.array_equal(sample.cpu().numpy()[0], rs)
        assert torch.equal(engine.decode_blocks(params, lay, q[2], q[3], 11, K, idx), sample)


@pytest.mark.parametrize("n,bs,B,eps1,n_t", [(8192, None, 20, 1.2, 1), (8192, None, 10, 1.2, 3), (8192, 3000, 20, 1.2, 2), (5000, None, 30, 1.0, 2),
                                              (8192, 2048, 20, 1.2, 1), (20000, None, 20, 1.2, 1),
                                              # the one-team gang builds: beam counts / sample counts without a three-team build
                                              (8192, None, 32, 1.2, 1), (3000, None, 50, 1.0, 2), (4096, None, 60, 1.2, 1), (3000, None, 10, 1.6, 1),
                                              (5000, None, 30, 1.34, 1)])
def test_gangs_of_teams_code_the_blocks_of_a_small_call(engine, oracle, n, bs, B, eps1, n_t):
    """Round 5: the reference's default `block_size=None` on ONE image's latents is one block of 8192 dims -- on one team of one CU
    40 ms, 255 CUs idle.  A call of fewer blocks than team slots is coded by GANGS (irec_team.hip): G teams per block, a chunk of 1024
    dims (or several) each, group sums exchanged through HBM and added in group order.  Same bits as the one-team form (NO_SPLIT) and as
    the oracle; ragged blocks (two table dims), beam passes (B = 30), sample stripes (one block: 8 chunk owners x 9 stripes).
    Gang builds: three teams per workgroup for B <= 30 where the LDS holds them, else one team (B = 32 ... 60, S = 122, B = 30 at S = 56)."""
    import irec
    S = oracle.n_samples(3.0, eps1)
    stats = [oracle.synthetic_latent(9300 + i, n) for i in range(n_t)]
    q = tuple(torch.from_numpy(np.stack([st[k] for st in stats])).cuda().contiguous() for k in range(4))
    lay = engine.layout(n_t, n, bs, 42)
    params = engine.params(3.0, S, B)
    alone = engine.params(3.0, S, B, irec._lib.IREC_FLAG_NO_SPLIT)
    plan = engine.plan(params, lay, 256)
    assert plan["kernel"].endswith(",gang>") and plan["split"] >= 2, plan
    assert plan["grid"] * plan["teams_per_wg"] >= lay.n_blocks * plan["split"], plan      # every member in the static round
    assert not engine.plan(alone, lay, 256)["kernel"].endswith(",gang>")
    K, idx, sample = engine.encode_blocks(params, lay, *q, 42, 256)
    K1, idx1, sample1 = engine.encode_blocks(alone, lay, *q, 42, 256)
    Kh = K.cpu().numpy()
    assert Kh.min() >= 1 and torch.equal(K, K1) and torch.equal(sample, sample1)
    ih, ih1 = idx.cpu().numpy(), idx1.cpu().numpy()
    for r in range(lay.n_blocks):
        assert np.array_equal(ih[r, :Kh[r]], ih1[r, :Kh[r]]), r
    assert torch.equal(engine.decode_blocks(params, lay, q[2], q[3], 42, K, idx), sample)
    bpt = lay.blocks_per_tensor
    ridx, rs = oracle.encode_tensor(*stats[n_t - 1], 42, 3.0, S, B, block_size=bs)
    got = [ih[lay.natural[(n_t - 1) * bpt + j], :Kh[lay.natural[(n_t - 1) * bpt + j]]].tolist() for j in range(bpt)]
    assert (got[0] == ridx) if bs is None else (got == ridx)
    assert np.array_equal(sample[n_t - 1].cpu().numpy(), rs)


def test_gangs_whose_members_own_several_chunks_and_share_cus(engine, oracle):
    """More blocks x chunks than team slots: 200 blocks of 5 chunks on 768 slots -- three chunk owners per block (two chunks, two, one), three
    members per CU, no stripes; a stripe cap in the flags changes nothing there (stripes only where CUs are idle).  Against the one-team form
    on every block, against the oracle on two.  Then 12 of the blocks: 60 chunk owners, four stripes each on 256 CUs."""
    import irec
    n, n_t, S, B = 5000, 200, 36, 20
    stats = [oracle.synthetic_latent(9500 + i, n) for i in range(n_t)]
    q = tuple(torch.from_numpy(np.stack([st[k] for st in stats])).cuda().contiguous() for k in range(4))
    lay = engine.layout(n_t, n, None, 42)
    alone = engine.params(3.0, S, B, irec._lib.IREC_FLAG_NO_SPLIT)
    K1, idx1, sample1 = engine.encode_blocks(alone, lay, *q, 42, 64)
    Kh, ih1 = K1.cpu().numpy(), idx1.cpu().numpy()
    for flags in (0, 2 << 12):
        params = engine.params(3.0, S, B, flags)
        plan = engine.plan(params, lay, 64)
        assert plan["kernel"] == "encode_chunk_kernel<20,10,3,gang>" and plan["split"] == 3 and plan["grid"] == plan["n_cu"], plan
        K, idx, sample = engine.encode_blocks(params, lay, *q, 42, 64)
        assert torch.equal(K, K1) and torch.equal(sample, sample1)
        ih = idx.cpu().numpy()
        assert all(np.array_equal(ih[r, :Kh[r]], ih1[r, :Kh[r]]) for r in range(lay.n_blocks))
    for i in (0, n_t - 1):
        ridx, rs = oracle.encode_tensor(*stats[i], 42, 3.0, S, B, block_size=None)
        assert ih1[lay.natural[i], :Kh[lay.natural[i]]].tolist() == ridx and np.array_equal(sample1[i].cpu().numpy(), rs)
    lay12 = engine.layout(12, n, None, 42)
    q12 = tuple(t[:12].contiguous() for t in q)
    plan = engine.plan(engine.params(3.0, S, B), lay12, 64)
    assert plan["split"] == 5 * (plan["n_cu"] // 60), plan
    K, idx, sample = engine.encode_blocks(engine.params(3.0, S, B), lay12, *q12, 42, 64)
    K12, i12 = K.cpu().numpy(), idx.cpu().numpy()
    assert torch.equal(sample, sample1[:12])
    for i in range(12):
        r, r1 = lay12.natural[i], lay.natural[i]
        assert K12[r] == Kh[r1] and np.array_equal(i12[r, :K12[r]], ih1[r1, :Kh[r1]]), i


def test_gang_calls_under_contention_match_the_one_team_form(engine, monkeypatch):
    """scripts/soak_gangs_threads.py in small: three threads issue gang calls of random shapes back to back on their own streams, so the
    members of a gang run late and out of step with each other.  Round 5 found its one gang bug here (11 of 360 calls differed): the sums a
    member hands over must have DRAINED (s_waitcnt vmcnt(0) in every wave) before its arrival is counted -- the team barrier's
    workgroup-scoped release fence does not wait for vector-memory stores.  Every call against the same call on one team per block."""
    import runpy
    monkeypatch.setenv("SOAK_THREADS", "3")
    monkeypatch.setenv("SOAK_CALLS", "120")
    monkeypatch.setenv("SOAK_SEED", "3")
    with pytest.raises(SystemExit) as e:
        runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "soak_gangs_threads.py"), run_name="__main__")
    assert e.value.code == 0


def test_ten_beam_encoder_random_shapes_against_the_oracle(engine, monkeypatch):
    """scripts/soak_parity.py SOAK_TEN=1 in small (round 6): random blocks of 1 .. 1024 dims, 2 .. 10 beams, S * 10 <= 256, 1 .. 300 blocks per
    call, benign to extreme statistics, K up to 300 (beyond the table window: the second pass), every encoder variant against the oracle --
    the team variant runs encode_ten_kernel.  The long run: profiles/r06w/soak_ten.log (31 709 blocks, 0 mismatches)."""
    import runpy
    monkeypatch.setenv("SOAK_TEN", "1")
    monkeypatch.setenv("SOAK_CASES", "70")
    monkeypatch.setenv("SOAK_SEED", "23")
    with pytest.raises(SystemExit) as e:
        runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "soak_parity.py"), run_name="__main__")
    assert e.value.code == 0


def test_split_encoder_and_shared_rows_under_contention_match_the_unshared_form(engine, monkeypatch):
    """The same stress for the other two cooperative forms (round 5's review, Next #3c: only the gang form had a contention test): three
    threads issue calls of blocks of at most 1024 dims back to back on their own streams -- calls of fewer than 64 blocks take the split
    encoder (workgroups of a block exchange tagged 8-byte granules), calls of 64 blocks up to 1.5 per CU share rows between teams (the same
    granules) -- so partners run late and out of step.  Every call against the same call with IREC_FLAG_NO_SPLIT; a call that gave up
    (partners not resident) is counted, not compared."""
    import runpy
    monkeypatch.setenv("SOAK_THREADS", "3")
    monkeypatch.setenv("SOAK_CALLS", "100")
    monkeypatch.setenv("SOAK_SEED", "5")
    monkeypatch.setenv("SOAK_MODE", "small")
    with pytest.raises(SystemExit) as e:
        runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "soak_gangs_threads.py"), run_name="__main__")
    assert e.value.code == 0


def test_a_gang_really_short_of_cus_gives_up_and_is_coded_again(engine, oracle):
    """No test hook: another stream holds 200 CUs for seconds (200 blocks of 65 536 dims, one team each) while ONE block of 16 384 dims asks
    for a gang of 144 teams, one per CU.  The members that find a CU wait 100 ms, poison the block's counter and leave; the rest start when
    CUs come free, see the poison and leave; BeamSearchCoder codes the call again on one team -- the bits of the one-team form.
    (Many SMALL gangs next to such a hog need no give-up: gangs drain in workgroup order, test_two_gang_calls_in_flight_on_two_streams.)"""
    import irec
    n, S, B = 16384, 36, 20
    st = oracle.synthetic_latent(9900, n)
    q = tuple(torch.from_numpy(a[None]).cuda().contiguous() for a in st)
    lay = engine.layout(1, n, None, 42)
    assert engine.plan(engine.params(3.0, S, B), lay, 256)["split"] == 144                    # (16 chunk owners x 9 sample stripes)
    K1, idx1, sample1 = engine.encode_blocks(engine.params(3.0, S, B, irec._lib.IREC_FLAG_NO_SPLIT), lay, *q, 42, 256)
    k1 = int(K1.cpu()[0]); want = idx1.cpu().numpy()[0, :k1].tolist()
    big = oracle.synthetic_latent(9950, 65536)
    qb = tuple(torch.from_numpy(np.stack([a] * 200)).cuda().contiguous() for a in big)
    layb = engine.layout(200, 65536, None, 7)
    hog = engine.params(3.0, S, B, irec._lib.IREC_FLAG_NO_SPLIT)
    assert engine.plan(hog, layb, 1024)["grid"] == 200
    # (HIP multiplexes its streams onto a few hardware queues: a side stream that lands on the queue of the current stream runs the two calls
    #  one after the other -- no shortage of CUs, nothing to give up, 2.6 s until the call is coded.  Which queue a new stream gets depends on
    #  how many the process has made before, i.e. on the tests that ran earlier: up to four side streams are tried.)
    gave_up = False
    for attempt in range(4):
        c = irec.BeamSearchCoder(kl_per_partition=3., n_beams=B, extra_samples=1.2, block_size=None)
        c._max_K_hint = 256
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            Kb, _, _ = engine.encode_blocks(hog, layb, *qb, 7, 1024)            # ~2.6 s on 200 CUs
        time.sleep(0.2)                                                          # (the long call is running)
        t0 = time.perf_counter()
        idx, sample = c.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42)
        t_gang = time.perf_counter() - t0
        print(f"[gang short of CUs] attempt {attempt}: give-ups {c._split_strikes}, {t_gang:.2f} s until the call was coded")
        assert [int(v) for v in idx] == want and torch.equal(sample, sample1)
        torch.cuda.synchronize()
        assert int(Kb.cpu().min()) > 100                                         # the long call coded its blocks meanwhile
        if c._split_strikes == 1 and t_gang > 0.1:                               # it did give up, and was coded again without sharing
            gave_up = True
            break
        assert c._split_strikes == 0 and t_gang > 1.0, (c._split_strikes, t_gang)   # (else: the two calls never met on the device)
    assert gave_up


def test_gang_calls_inside_a_replayed_graph(engine, oracle):
    """A gang call is plain stream work (its arrival counters are zeroed by the call's own preparation kernel), so a captured sequence of
    such calls -- irec.models.GraphedCompress with block_size=None -- stays correct on every replay."""
    stats = [oracle.synthetic_latent(9800 + i, 4096) for i in range(2)]
    t = [torch.from_numpy(np.stack([st[k] for st in stats])).cuda().contiguous() for k in range(4)]
    refs = [oracle.encode_tensor(*stats[i], 42, 3.0, 36, 20, block_size=None) for i in range(2)]
    c = _coder(3.0, 20, 1.2, block_size=None, variant="auto")
    c.encode_tensors_device(*t, 42, None, max_K=64).to_lists()             # (warm: the partition hint and the stream's scratch)
    lay = engine.layout(2, 4096, None, 42)
    assert engine.plan(c._params(), lay, 64)["kernel"].endswith(",gang>")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        c.encode_tensors_device(*t, 42, None, max_K=64)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        p1 = c.encode_tensors_device(*t, 42, None, max_K=64)
        p2 = c.encode_tensors_device(*t, 42, None, max_K=64)
    for rep in range(3):
        g.replay()
        torch.cuda.synchronize()
        for p in (p1, p2):
            lists = p.to_lists()
            for i in range(2):
                got = lists[i][0] if isinstance(lists[i][0], (list, tuple)) else lists[i]
                assert [int(v) for v in got] == refs[i][0] and np.array_equal(p.sample[i].cpu().numpy(), refs[i][1]), (rep, i)


def test_gang_blocks_with_nothing_to_code_and_too_small_an_index_buffer(engine, oracle):
    """A gang whose block has KL = 0 (posterior == prior: K = 0, sample = p.loc, written by the chunk owners' first stripes) next to ordinary
    blocks; and max_K below the blocks' K: every member leaves the block uncoded with out_K = K (what BeamSearchCoder raises its hint from)."""
    n, n_t, S, B = 5000, 3, 36, 20
    stats = [oracle.synthetic_latent(9700 + i, n) for i in range(n_t)]
    ql, qs, pl, ps = (torch.from_numpy(np.stack([st[k] for st in stats])).cuda().contiguous() for k in range(4))
    ql[1] = pl[1]; qs[1] = ps[1]
    lay = engine.layout(n_t, n, None, 42)
    params = engine.params(3.0, S, B)
    assert engine.plan(params, lay, 64)["kernel"].endswith(",gang>")
    K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, 64)
    Kh, ih = K.cpu().numpy(), idx.cpu().numpy()
    assert Kh[lay.natural[1]] == 0 and torch.equal(sample[1], pl[1])
    for i in (0, 2):
        ridx, rs = oracle.encode_tensor(*stats[i], 42, 3.0, S, B, block_size=None)
        assert ih[lay.natural[i], :Kh[lay.natural[i]]].tolist() == ridx and np.array_equal(sample[i].cpu().numpy(), rs)
    K8, _, _ = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, 8)        # K ~ 38 > 8
    assert np.array_equal(K8.cpu().numpy(), Kh)


def test_two_gang_calls_in_flight_on_two_streams(engine, oracle):
    """Two calls of 24 one-block latents each (192 members each, one per CU) issued from two threads on two streams: 384 workgroups do not
    fit 256 CUs, so the later call's members wait for the earlier call's to leave -- 5 ms, far from the 100 ms give-up -- or, if a call does
    give up, BeamSearchCoder codes it again on one team per block.  Either way: no hang, the oracle's bits on both streams, every repetition."""
    import threading
    import irec
    n_t, n, S, B = 24, 8192, 36, 20
    stats = [[oracle.synthetic_latent(9600 + 100 * w + i, n) for i in range(n_t)] for w in range(2)]
    qs = [tuple(torch.from_numpy(np.stack([st[k] for st in stats[w]])).cuda().contiguous() for k in range(4)) for w in range(2)]
    refs = [oracle.encode_tensor(*stats[w][n_t - 1], 42, 3.0, S, B, block_size=None) for w in range(2)]
    torch.cuda.synchronize()
    results, errors = [[], []], []

    def work(w):
        try:
            c = irec.BeamSearchCoder(kl_per_partition=3., n_beams=B, extra_samples=1.2, block_size=None)
            with torch.cuda.stream(torch.cuda.Stream()):
                for rep in range(4):
                    idx, sample = c.encode(_normal(qs[w][0], qs[w][1]), _normal(qs[w][2], qs[w][3]), seed=42, batched=True)
                    results[w].append((idx[n_t - 1], sample[n_t - 1].cpu().numpy(), c._split_strikes))
        except Exception as e:                      # noqa: BLE001 (reported below, in the main thread)
            errors.append((w, repr(e)))
    th = [threading.Thread(target=work, args=(w,)) for w in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a gang call hangs"
    assert not errors, errors
    assert time.perf_counter() - t0 < 60
    for w in range(2):
        assert len(results[w]) == 4
        for idx, sample, _ in results[w]:
            assert [int(v) for v in idx] == refs[w][0] and np.array_equal(sample, refs[w][1]), w


def test_a_gang_whose_partners_are_not_resident_gives_up(engine, oracle):
    """IREC_FLAG_TEST_SPLIT_ORPHAN: every member but the first leaves at once; member 0 waits its 100 ms, poisons the block's arrival counter
    and reports the block as not coded (-2); BeamSearchCoder codes the call again on one team (the back-off of test_recovery_from_a_give_up)."""
    import irec
    stats = oracle.synthetic_latent(9400, 4096)
    q = tuple(torch.as_tensor(a[None], device="cuda") for a in stats)
    lay = engine.layout(1, 4096, None, 42)
    orphan = engine.params(3.0, 36, 20, irec._lib.IREC_FLAG_TEST_SPLIT_ORPHAN)
    assert engine.plan(orphan, lay, 64)["split"] >= 4                              # (4 chunk owners x sample stripes)
    t0 = time.perf_counter()
    K, _, _ = engine.encode_blocks(orphan, lay, *q, 42, 64)
    assert int(K.cpu()[0]) == -2 and 0.09 < time.perf_counter() - t0 < 1.0
    ridx, rs = oracle.encode_tensor(*stats, 42, 3.0, 36, 20, block_size=None)
    c = irec.BeamSearchCoder(kl_per_partition=3., n_beams=20, extra_samples=1.2, block_size=None)
    c._test_split_orphan = True
    idx, sample = c.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42)
    assert [int(v) for v in idx] == ridx and np.array_equal(sample.cpu().numpy()[0], rs)
    assert (c._split_strikes, c._split_pause) == (1, 0)
    c._test_split_orphan = False
    idx, sample = c.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42)     # shared again, whole: count reset
    assert [int(v) for v in idx] == ridx and np.array_equal(sample.cpu().numpy()[0], rs) and c._split_strikes == 0


def test_wide_beam_uses_generic_path(engine, oracle):
    mq, sq, mp, sp = oracle.synthetic_latent(78, 300)
    ridx, rs = oracle.encode_block(mq, sq, mp, sp, 9, 2.0, 7, 50)
    for generic in (False, True):    # B = 50 > 32, S = 7 < B: the 60-beam team build (round 3), and the generic kernel pinned
        c = _coder(2.0, 50, 1.0)
        c.force_generic = generic
        idx, sample = c.encode(_normal(mq[None], sq[None]), _normal(mp[None], sp[None]), seed=9)
        assert [int(i) for i in idx] == ridx and np.array_equal(sample.cpu().numpy()[0], rs), generic
    ridx, rs = oracle.encode_block(mq, sq, mp, sp, 9, 2.0, 7, 64)
    c = _coder(2.0, 64, 1.0)         # 60 < B <= 64: generic kernel
    idx, sample = c.encode(_normal(mq[None], sq[None]), _normal(mp[None], sp[None]), seed=9)
    assert [int(i) for i in idx] == ridx and np.array_equal(sample.cpu().numpy()[0], rs)


@pytest.mark.parametrize("B,omega,eps1,n", [(100, 3.0, 1.2, 900), (256, 2.0, 1.0, 700), (65, 5.0, 1.0, 700), (200, 3.0, 1.0, 1500)])
def test_more_than_64_beams_take_the_generic_kernel(engine, oracle, B, omega, eps1, n):
    """n_beams is any Python int in the reference (beam_search_coder.py:28); round 4 lifts this build's limit from 64 to 256 (the
    generic kernel: beams in its slab, parents in 8 bits of the back-pointers, selection by the scan once more than 64 survive the
    threshold).  S < B on the first steps (S = 7 at Omega = 2: 7, 49, 256 beams), blocks of more than 1024 dims, decode(encode) exact."""
    S = oracle.n_samples(omega, eps1)
    mq, sq, mp, sp = oracle.synthetic_latent(600 + B, n)
    ridx, rs = oracle.encode_block(mq, sq, mp, sp, 11, omega, S, B)
    c = _coder(omega, B, eps1)
    idx, sample = c.encode(_normal(mq[None], sq[None]), _normal(mp[None], sp[None]), seed=11)
    assert [int(i) for i in idx] == ridx and np.array_equal(sample.cpu().numpy()[0], rs)
    assert torch.equal(c.decode(_normal(mp[None], sp[None]), idx, seed=11), sample)
    lay = engine.layout(1, n, None, 11)
    assert engine.plan(engine.params(omega, S, B), lay, 32)["kernel"] == "encode_generic_kernel"


@pytest.mark.parametrize("omega,B", [(4.18, 65), (4.18, 100), (4.2, 70)])
def test_first_step_of_more_beams_than_lane_maxima_guarantee(engine, oracle, omega, B):
    """Round 4's advice: with more than 64 beams the threshold of the selection is the SMALLEST lane maximum, which guarantees 64
    survivors, not Bnew.  At step 0 with S = 65 (66) candidates and B >= S, Bnew = S: when the 64 largest keys sit in 64 distinct lanes
    (2 in 65 blocks at S = 65) only 64 survive and rank_survivors used to record ranks 0..63 and return -- beams 64.. kept stale
    parents.  Now it sends the shortfall to the scan.  Many blocks, K >= 2 so that step 0's selection feeds a second step."""
    S = oracle.n_samples(omega, 1.0)
    assert S in (65, 66)
    n_t, n = 96, 420
    stats = [oracle.synthetic_latent(7000 + i, n) for i in range(n_t)]
    ql, qs, pl, ps = (np.stack([s[k] for s in stats]) for k in range(4))
    c = _coder(omega, B, 1.0)
    idx, sample = c.encode(_normal(ql, qs), _normal(pl, ps), seed=5, batched=True)
    deep = 0
    for t in range(n_t):
        ridx, rs = oracle.encode_block(ql[t], qs[t], pl[t], ps[t], 5, omega, S, B)
        assert [int(i) for i in idx[t]] == ridx, t
        assert np.array_equal(sample[t].cpu().numpy(), rs), t
        deep += len(ridx) >= 2
    assert deep >= n_t // 2
    lay = engine.layout(n_t, n, None, 5)
    assert engine.plan(engine.params(omega, S, B), lay, 32)["kernel"] == "encode_generic_kernel"


def test_wide_beam_blocks_beyond_the_table_window_take_the_generic_kernel(engine, oracle):
    """B = 50 has no fused-Philox fast encoder: blocks with more partitions than the proposal tables cover are coded by the
    generic kernel in the call's second pass (irec_host.cpp: team_only plans) -- same outputs as the oracle's."""
    n, bs = 2192, 1000
    stats = oracle.synthetic_latent(515, n)
    ql, qs, pl, ps = (torch.from_numpy(a[None]).cuda().contiguous() for a in stats)
    lay = engine.layout(1, n, bs, 42)
    ridx, rs = oracle.encode_tensor(*stats, 42, 3.0, 20, 50, block_size=bs)
    assert max(len(i) for i in ridx) >= 6 and min(len(i) for i in ridx) <= 3       # two long blocks, one short
    params = engine.params(3.0, 20, 50, table_steps=3)
    K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, 32)
    Kh, ih = K.cpu().numpy(), idx.cpu().numpy()
    got = [ih[lay.natural[j], :Kh[lay.natural[j]]].tolist() for j in range(3)]
    assert engine.plan(params, lay, 32)["table_steps"] == 3 and got == ridx
    assert np.array_equal(sample.cpu().numpy()[0], rs)


@pytest.mark.parametrize("variant", ["table", "fused"])
def test_stress_config_multi_pass_eight_wave_path(engine, oracle, variant):
    # BASELINE config 5 with eps = 0.2: B = 30, S = 403 -> 12090 candidates per step; the partial scores no longer fit
    # LDS in one go (sample passes) and only one workgroup fits per CU (8-wave variant of the fast encoder)
    mq, sq, mp, sp = oracle.synthetic_latent(4242, 1000)
    c = _coder(5.0, 30, 1.2, variant=variant)
    assert c.n_samples == 403
    idx, sample = c.encode(_normal(mq[None], sq[None]), _normal(mp[None], sp[None]), seed=42)
    ridx, rs = oracle.encode_block(mq, sq, mp, sp, 42, 5.0, 403, 30)
    assert [int(i) for i in idx] == ridx and np.array_equal(sample.cpu().numpy()[0], rs)
    assert torch.equal(c.decode(_normal(mp[None], sp[None]), idx, seed=42), sample)


@pytest.mark.parametrize("D", [5, 65, 130, 250])
def test_ragged_dims_with_idle_lanes_many_partitions(engine, oracle, D):
    # found by scripts/soak_parity.py: with the proposal table, lanes beyond the padded row end must not read past the
    # row (0 * garbage-NaN poisoned the scores).  Tight posteriors -> K ~ 100 steps, S = 2 < B = 27.
    rng = np.random.default_rng(D)
    mp = rng.normal(0, 1, D); sp = np.exp(rng.normal(0, 0.5, D))
    mq = mp + sp * rng.normal(0, 1.0, D); sq = sp * rng.uniform(0.05, 0.5, D)
    mq, sq, mp, sp = (a.astype(np.float32) for a in (mq, sq, mp, sp))
    for variant in VARIANTS:
        c = _coder(1.0, 27, 1.0, variant=variant)
        assert c.n_samples == 2
        idx, sample = c.encode(_normal(mq[None], sq[None]), _normal(mp[None], sp[None]), seed=949676964)
        ridx, rs = oracle.encode_block(mq, sq, mp, sp, 949676964, 1.0, 2, 27, max_K=2048)
        assert [int(i) for i in idx] == ridx, variant
        assert np.array_equal(sample.cpu().numpy()[0], rs), variant


@pytest.mark.parametrize("D", [1, 5, 65, 130, 250, 257, 600, 768, 1023, 1024])
def test_one_beam_ragged_dims_and_tiny_sample_counts(engine, oracle, D):
    """encode_lone_kernel on blocks of one to four dim groups with idle lanes, idle groups (600 dims: the fourth group of a
    pair is all zero coefficients) and S = 1, 2, 3, 5, 9 -- fewer samples than a reduce-scatter holds, a sample count that is
    not a multiple of four --, tight posteriors (tens of steps): indices and sample against the oracle."""
    rng = np.random.default_rng(1000 + D)
    mp = rng.normal(0, 1, D); sp = np.exp(rng.normal(0, 0.5, D))
    mq = mp + sp * rng.normal(0, 0.7, D); sq = sp * rng.uniform(0.2, 0.7, D)
    mq, sq, mp, sp = (a.astype(np.float32) for a in (mq, sq, mp, sp))
    for omega, S in ((0.5, 1), (1.0, 2), (1.2, 3), (1.7, 5), (2.2, 9)):
        c = _coder(omega, 1, 1.0, variant="table")
        c.table_steps = 2048                      # (a window that covers every K here: nothing is left to the second pass)
        assert c.n_samples == S
        lay = engine.layout(1, D, None, 7)
        plan = engine.plan(c._params(), lay, 2048)
        assert plan["kernel"] == "encode_lone_kernel" and plan["table_steps"] == 2048, plan
        idx, sample = c.encode(_normal(mq[None], sq[None]), _normal(mp[None], sp[None]), seed=7)
        ridx, rs = oracle.encode_block(mq, sq, mp, sp, 7, omega, S, 1, max_K=65536)
        assert [int(i) for i in idx] == ridx, (D, S)
        assert np.array_equal(sample.cpu().numpy()[0], rs), (D, S)
        assert torch.equal(c.decode(_normal(mp[None], sp[None]), idx, seed=7).cpu(), sample.cpu())


@pytest.mark.parametrize("B,n_tensors", [(1, 383), (1, 384), (1, 385), (1, 421), (20, 95), (20, 97), (20, 131), (10, 104)])
def test_block_hand_out_codes_every_block_once(engine, oracle, B, n_tensors):
    """The XCD-aware hand-out of the batch encoders (irec_fast_common.h: xcd_static_row / xcd_pull_row): calls of many tiny
    blocks (tensors of 64 dims in 8 blocks of 8) whose block counts sit on either side of what the static first round deals
    (12 x n_CU one-beam blocks, 3 x n_CU team blocks), are not multiples of 64 and leave the per-XCD counters ragged shares --
    every tensor against the oracle, so a block coded twice, skipped or coded from another block's row would show."""
    n, bs = 64, 8
    q = [np.stack([oracle.synthetic_latent(4000 + i, n)[j] for i in range(n_tensors)]) for j in range(4)]
    c = _coder(2.0, B, 1.0, block_size=bs)
    S = oracle.n_samples(2.0, 1.0)
    lay = engine.layout(n_tensors, n, bs, 42)
    plan = engine.plan(c._params(), lay, 32)
    assert plan["kernel"].startswith("encode_lone_kernel" if B == 1 else ("encode_ten_kernel", "encode_team_kernel")), plan
    assert lay.n_blocks == 8 * n_tensors and plan["grid"] % 8 == 0
    idx, sample = c.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42, batched=True)
    for i in range(n_tensors):
        ridx, rs = oracle.encode_tensor(q[0][i], q[1][i], q[2][i], q[3][i], 42, 2.0, S, B, block_size=bs)
        assert idx[i] == ridx and np.array_equal(sample[i].cpu().numpy(), rs), i


def test_zero_kl_block(engine):
    mp = torch.tensor([[0.3, -1.0, 2.0]]); sp = torch.tensor([[1.0, 2.0, 0.5]])
    c = _coder(3.0, 10, 1.0)
    idx, sample = c.encode(_normal(mp, sp), _normal(mp, sp), seed=1)
    assert idx == [] and torch.equal(sample.cpu(), mp)
    assert torch.equal(c.decode(_normal(mp, sp), [], seed=1).cpu(), mp)


def test_max_k_retry(engine, oracle):
    mq, sq, mp, sp = oracle.synthetic_latent(5, 1000)
    c = _coder(3.0, 20, 1.2)
    c._max_K_hint = 1
    idx, sample = c.encode(_normal(mq[None], sq[None]), _normal(mp[None], sp[None]), seed=42)
    ridx, rs = oracle.encode_block(mq, sq, mp, sp, 42, 3.0, 36, 20)
    assert [int(i) for i in idx] == ridx and c._max_K_hint >= len(ridx)


def test_high_kl_block_many_partitions(engine, oracle):
    mq, sq, mp, sp = oracle.synthetic_latent(6, 256)
    sq = (sq * 0.2).astype(np.float32)  # KL ~ 1.2 nats/dim -> K ~ 100
    c = _coder(3.0, 10, 1.0)
    idx, sample = c.encode(_normal(mq[None], sq[None]), _normal(mp[None], sp[None]), seed=8)
    ridx, rs = oracle.encode_block(mq, sq, mp, sp, 8, 3.0, 20, 10)
    assert len(ridx) > 64 and [int(i) for i in idx] == ridx and np.array_equal(sample.cpu().numpy()[0], rs)


def test_decoder_both_table_paths(engine, oracle):
    """The decoder gathers the quantile table through the L2 for small calls and from an LDS copy from 16 blocks per CU on
    (4608 blocks here): both must reproduce the encoder's sample, and the oracle's decode of the same indices."""
    n_t, n, bs = 512, 8192, 1000
    rng = np.random.default_rng(77)
    mp = rng.normal(0, 1, (n_t, n)).astype(np.float32); lsp = rng.normal(0, 0.25, (n_t, n)).astype(np.float32)
    sp = np.exp(lsp).astype(np.float32)
    mq = (mp + sp * rng.normal(0, 0.2, (n_t, n))).astype(np.float32)
    sq = np.exp(lsp - np.abs(rng.normal(0, 0.05, (n_t, n)))).astype(np.float32)
    ql, qs, pl, ps = (torch.from_numpy(a).cuda().contiguous() for a in (mq, sq, mp, sp))
    lay = engine.layout(n_t, n, bs, 42)
    params = engine.params(3.0, 36, 20, 0)
    K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, 32)
    assert lay.n_blocks >= 16 * engine.plan(params, lay, 32)["n_cu"]
    rec = engine.decode_blocks(params, lay, pl, ps, 42, K, idx)                        # LDS-table decoder
    assert torch.equal(rec, sample)
    few = engine.layout(3, n, bs, 42)                                                   # 27 blocks: the L2 path
    rows = torch.as_tensor(np.concatenate([lay.natural[i * 9:(i + 1) * 9] for i in (0, 255, 511)]))
    sel = torch.as_tensor([0, 255, 511])
    K3, idx3 = K.cpu()[rows], idx.cpu()[rows]
    order = torch.as_tensor(np.argsort(few.natural))                                    # rows of `few` in its own layout order
    rec3 = engine.decode_blocks(params, few, pl[sel.cuda()].contiguous(), ps[sel.cuda()].contiguous(), 42,
                                K3[order].cuda().contiguous(), idx3[order].cuda().contiguous())
    assert torch.equal(rec3, sample[sel.cuda()])
    Kh, ih = K.cpu().numpy(), idx.cpu().numpy()
    for i in (0, 511):
        blocks = [ih[lay.natural[i * 9 + j], :Kh[lay.natural[i * 9 + j]]].tolist() for j in range(9)]
        assert np.array_equal(oracle.decode_tensor(mp[i], sp[i], blocks, 42, 36, block_size=bs), sample[i].cpu().numpy())


def test_randomised_soak_every_variant_against_the_oracle():
    """scripts/soak_parity.py in the driver's run (the soaks were builder-run only): random block shapes (1 to 1024
    dims), beams 1 to 64, S from 2 to several hundred, five statistics regimes incl. K up to 300, 1 to 8 blocks per call --
    the team, one-table, fused-Philox and generic encoders and the decoder against the CPU oracle, bit for bit; then the same
    biased to more than 1024 candidates per step (sample passes, keys in the slab, streamed top-B)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({"SOAK_CASES": "150", "SOAK_SEED": "31"}, {"SOAK_CASES": "40", "SOAK_SEED": "32", "SOAK_BIG": "1"}):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "soak_parity.py")], env=dict(os.environ, **extra),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "mismatches: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ---- the reference's own hyper-parameter sweep (examples/lossless/data_aggregation.py:5-7) -------------------------------
SWEEP_OMEGA = (2, 3, 4, 5, 6)
SWEEP_EPS1 = (1.0, 1.1, 1.2, 1.5)
SWEEP_BEAMS = (1, 10, 50)


@pytest.mark.parametrize("B", SWEEP_BEAMS)
@pytest.mark.parametrize("omega", SWEEP_OMEGA)
def test_reference_sweep_grid(engine, oracle, omega, B):
    """All 60 cells of the grid the reference sweeps -- kl_per_partition in 2..6 x extra_samples in {1, 1.1, 1.2, 1.5} x
    n_beams in {1, 10, 50}, S = int(exp(Omega * (1 + eps))) from 7 to 8103 (beam_search_coder.py:28-29) -- on one 1000-dim and
    one 192-dim block each: emitted indices and sample bit-exact against the oracle, once the way the library picks its
    kernels for a call this small and once with the batch encoder pinned (what a batch of such blocks runs on:
    encode_lone_kernel for B = 1, encode_team_kernel<10,3,1[,passes]> for B = 10, <60,1,3> / <54,1,3> (S >= 128) for B = 50; S * B reaches 405 150
    candidates per step)."""
    from irec import _lib
    n, bs = 1192, 1000
    stats = oracle.synthetic_latent(4242, n)
    ql, qs, pl, ps = (torch.from_numpy(a[None]).cuda().contiguous() for a in stats)
    lay = engine.layout(1, n, bs, 42)
    for eps1 in SWEEP_EPS1:
        S = oracle.n_samples(float(omega), eps1)
        ridx, rs = oracle.encode_tensor(*stats, 42, float(omega), S, B, block_size=bs)
        assert max(len(i) for i in ridx) <= 32
        # (one beam: the pinned call runs encode_lone_kernel, one wave per block; pinning the team shape as well keeps the
        #  team encoder's one-beam builds <10,3,1[,passes],one> under test)
        for flags in (0, _lib.IREC_FLAG_TEAM) + ((_lib.IREC_FLAG_TEAM | _lib.IREC_FLAG_SHAPE["team"],) if B == 1 else ()):
            params = engine.params(float(omega), S, B, flags)
            K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, 32)
            Kh, ih = K.cpu().numpy(), idx.cpu().numpy()
            got = [ih[lay.natural[j], :Kh[lay.natural[j]]].tolist() for j in range(2)]
            plan = engine.plan(params, lay, 32)
            assert got == ridx, (omega, eps1, S, B, flags, plan["kernel"])
            assert np.array_equal(sample.cpu().numpy()[0], rs), (omega, eps1, S, B, flags, plan["kernel"])
            if flags and B <= 60:
                lone = B == 1 and not (flags & _lib.IREC_FLAG_SHAPE["team"])
                assert plan["kernel"].startswith("encode_lone_kernel" if lone else ("encode_ten_kernel", "encode_team_kernel")), plan
            assert torch.equal(engine.decode_blocks(params, lay, pl, ps, 42, K, idx), sample)


def test_one_beam_batch_one_wave_per_block(engine, oracle):
    """encode_lone_kernel (n_beams = 1: one wave per block) on a batch big enough that waves pull second and third blocks from
    the counter (3 300 blocks > 12 waves x 256 CUs), with blocks of one and four dim groups, zero-KL tensors
    (K = 0: sample = p.loc), a table window shorter than some blocks' K (those take the fused-Philox second pass) and, in a
    second call, a max_K below the largest K (out_K reports it, nothing else is written for the block): indices and
    samples of every tensor against the OpenMP build of the oracle, bit for bit; decode(encode) exact."""
    from irec import _lib
    n, bs, omega, eps1, N = 2192, 1000, 3.0, 1.0, 1100        # blocks of 1000, 1000 and 192 dims: four and one dim groups
    S = oracle.n_samples(omega, eps1)
    lat = [list(oracle.synthetic_latent(7000 + i, n)) for i in range(N)]
    for i in (5, 700):                                        # q == p: K = 0 everywhere
        lat[i][0] = lat[i][2].copy(); lat[i][1] = lat[i][3].copy()
    for i in (9, 933):                                        # a far-off posterior: K well beyond the others'
        lat[i][0] = (lat[i][2] + 2.5 * lat[i][3]).astype(np.float32)
    host = [np.stack([l[j] for l in lat]) for j in range(4)]
    dev = [torch.as_tensor(a, device="cuda") for a in host]
    lay = engine.layout(N, n, bs, 42)
    assert lay.n_blocks == 3 * N and lay.n_blocks > 12 * 256
    ridx, rsamp, _ = oracle.encode_tensors_omp(*host, 42, omega, S, 1, bs, max_K=4096)
    kmax = max(len(b) for t in ridx for b in t)
    kbulk = int(np.percentile([len(b) for t in ridx for b in t], 90))
    assert kmax > 2 * kbulk > 0
    for steps in (0, kbulk):                                  # the default window (covers max_K) / a window the outliers leave
        params = engine.params(omega, S, 1, 0, (), steps)
        plan = engine.plan(params, lay, kmax)
        assert plan["kernel"] == "encode_lone_kernel", plan
        K, idx, sample = engine.encode_blocks(params, lay, *dev, 42, kmax)
        Kh, ih = K.cpu().numpy(), idx.cpu().numpy()
        nat = np.asarray(lay.natural).reshape(N, -1)
        for i in range(N):
            got = [ih[r, :Kh[r]].tolist() for r in nat[i]]
            assert got == ridx[i], (steps, i)
        assert np.array_equal(sample.cpu().numpy(), rsamp), steps
        assert torch.equal(engine.decode_blocks(params, lay, dev[2], dev[3], 42, K, idx), sample)
    # max_K below the outliers' K: their rows report K and stay uncoded, every other block is coded as before
    params = engine.params(omega, S, 1)
    K2, idx2, _ = engine.encode_blocks(params, lay, *dev, 42, kbulk)
    K2h, i2h = K2.cpu().numpy(), idx2.cpu().numpy()
    assert np.array_equal(K2h, Kh) and (K2h > kbulk).any()
    ok = K2h <= kbulk
    assert all(i2h[r, :K2h[r]].tolist() == ih[r, :Kh[r]].tolist() for r in np.nonzero(ok)[0])


def test_scalar_parameters_out_of_range_are_errors(engine, oracle):
    """400 random combinations of kl_per_partition (0, negative, NaN, inf, denormal, huge), n_samples (0 .. 2^31 - 1), n_beams
    (0 .. 1000), flag words (every diagnostic bit), table_steps and max_K (negative .. 2^31 - 1) on a two-block call: each one
    either codes the blocks or comes back as an irec error with text -- no crash, no hang, no silent wrap."""
    import irec
    n = 1192
    stats = oracle.synthetic_latent(4242, n)
    ql, qs, pl, ps = (torch.from_numpy(a[None]).cuda().contiguous() for a in stats)
    lay = engine.layout(1, n, 1000, 42)
    rng = np.random.default_rng(3)
    ran = failed = 0
    for _ in range(400):
        om = float(rng.choice([0.0, -1.0, float("nan"), float("inf"), 1e-30, 1e30, 3.0]))
        S = int(rng.choice([0, -1, 1, 2, 36, 2 ** 24, 2 ** 24 + 1, 2 ** 31 - 1]))
        B = int(rng.choice([0, -3, 1, 2, 64, 65, 1000]))
        fl = int(rng.choice([0, 1, 2, 4, 8, 16, 128, 0xF00, 0x700, 1 << 12, 0xF << 12, 0x7FFFFFFF]))
        mk = int(rng.choice([-1, 0, 1, 32, 65536, 65537, 2 ** 31 - 1]))
        ts = int(rng.choice([0, -1, 1, 4096, 2 ** 31 - 1]))
        try:
            params = engine.params(om, S, B, fl, (), ts)
            if S > 200 or mk > 4096:                               # (sizes only: no launch of that size in a test)
                ws = engine.lib.irec_encode_workspace_bytes(engine.ctx, ctypes.byref(params), 1000, mk if 0 <= mk < 2 ** 31 else 0)
                assert ws >= 0
            else:
                K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, mk)
                torch.cuda.synchronize()
                assert K.shape[0] == lay.n_blocks
            ran += 1
        except (irec.CodingError, ValueError, AssertionError, OverflowError, ctypes.ArgumentError):
            failed += 1
    assert ran > 50 and failed > 50, (ran, failed)
    # and the library still codes afterwards
    params = engine.params(3.0, 36, 20)
    K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, 32)
    ridx, rs = oracle.encode_tensor(*stats, 42, 3.0, 36, 20, block_size=1000)
    Kh, ih = K.cpu().numpy(), idx.cpu().numpy()
    assert [ih[lay.natural[j], :Kh[lay.natural[j]]].tolist() for j in range(2)] == ridx


def test_many_beams_selection_refinement(engine):
    """top-B with B up to 64 (irec_fast_common.h: rank_survivors): the B-th largest of the 64 lane maxima leaves more than 64
    candidates above it, which are cut down to the exact B best by a bitwise search for the B-th largest key; against
    torch.sort (value descending, ties to the lower flat index), including heavy ties."""
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    for n, nsel, bcur, ties in [(3000, 50, 50, False), (1000, 64, 20, False), (777, 60, 7, False), (405150, 50, 50, False),
                                (3000, 50, 50, True), (12000, 33, 30, True), (60, 50, 10, False)]:
        sc = torch.randn(n, generator=g, device="cuda")
        if ties:
            sc = torch.round(sc * 8) / 8
        sel = engine.test_select(sc, nsel, bcur).cpu().numpy()
        order = torch.sort(sc.cpu().double() * 1.0, descending=True, stable=True).indices[:nsel].numpy()
        want = np.stack([order // bcur, order % bcur], axis=1)
        assert np.array_equal(sel, want), (n, nsel, bcur, ties)
        assert np.array_equal(engine.test_select(sc, nsel, bcur, quick=True).cpu().numpy(), want), (n, nsel, bcur, ties, "quick")


def test_new_entry_points_reject_what_they_cannot_serve(engine):
    """Round-3 entries of include/irec.h: sizes they cannot serve are irec_status errors with text, never a crash or a
    silent fallback -- tensors beyond the staged decoder, a decode scratch too small or misaligned, hand-off shapes that do
    not fit their operands."""
    from irec import _lib
    lib, ctx = engine.lib, engine.ctx
    params = engine.with_table_dims(engine.params(3.0, 36, 20), engine.layout(2, 8192, 1000, 42))
    assert lib.irec_decode_tensors_supported(ctypes.byref(params), 8192, 1000) == 1
    assert lib.irec_decode_tensors_supported(ctypes.byref(params), 301056, 1000) == 0       # does not fit the LDS
    assert lib.irec_decode_tensors_supported(ctypes.byref(params), 700, 7) == 0             # more units than a workgroup takes
    z = torch.zeros(64, dtype=torch.float32, device=engine.device)
    zi = torch.zeros(64, dtype=torch.int32, device=engine.device)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = lib.irec_beam_decode_tensors(ctx, ctypes.byref(params), 1, 301056, 1000, None, None, p(z), p(z), 42, 4, p(zi), p(zi), p(z), None, 0, None)
    assert st == _lib.IREC_E_INVALID and b"staged decoder" in lib.irec_last_error()
    lay = engine.layout(2, 8192, 1000, 42)
    need = lib.irec_decode_workspace_bytes(ctx, ctypes.byref(params), 8)
    assert need > 0
    ws = torch.empty(need + 512, dtype=torch.uint8, device=engine.device)
    big = torch.zeros(2 * 8192, dtype=torch.float32, device=engine.device)
    K = torch.zeros(lay.n_blocks, dtype=torch.int32, device=engine.device)
    idx = torch.zeros((lay.n_blocks, 8), dtype=torch.int32, device=engine.device)
    args = lambda w, nbytes: (ctx, ctypes.byref(params), lay.n_blocks, p(lay.block_base), p(lay.block_pos), p(lay.block_dim), lay.max_dim,
                              p(lay.perm), p(big), p(big), 42, 8, p(K), p(idx), p(big), w, nbytes, None)
    assert lib.irec_beam_decode_ws(*args(p(ws), need // 2)) == _lib.IREC_E_WORKSPACE
    assert lib.irec_beam_decode_ws(*args(ctypes.c_void_p(ws.data_ptr() + 4), need)) == _lib.IREC_E_WORKSPACE    # not 256-byte aligned
    assert lib.irec_beam_decode_ws(*args(p(ws), need)) == 0                                                      # K = 0 everywhere: sample = mu_p
    torch.cuda.synchronize()
    assert lib.irec_shim_stats(ctx, p(z), None, p(z), 4, 1, 8, 0, 2, 4, None, None, None) == _lib.IREC_E_INVALID        # four statistics need the inference heads
    assert lib.irec_shim_stats(ctx, p(z), p(z), p(z), 3, 1, 8, 4, 2, 4, None, None, None) == _lib.IREC_E_INVALID
    assert lib.irec_shim_cat_elu(ctx, p(z), None, p(z), 1, 4, 2, 4, 0, 4, None, None) == _lib.IREC_E_INVALID            # slice beyond the channels
    assert lib.irec_shim_residual_elu(ctx, p(z), p(z), 0.1, p(z), None, 1, 4, 4, None, None) == _lib.IREC_E_INVALID


def test_decoder_fast_sqrt_exhaustive(engine):
    """The decoder's 9-instruction square root (irec_decode.hip: dec_sqrt_core) returns sqrtf's bits -- the correctly rounded
    value the oracle's libm gives -- for EVERY float32 bit pattern it is allowed to see: all 2^32 patterns are run on the
    device (the excluded ones, non-zero values below 2^-96, take sqrtf itself)."""
    out = torch.zeros(2, dtype=torch.int64, device=engine.device)
    from irec import _lib
    _lib.check(engine.lib.irec_test_decoder_sqrt(engine.ctx, ctypes.c_void_p(out.data_ptr()), engine._stream()), "irec_test_decoder_sqrt")
    bad, seen = (int(v) for v in out.cpu())
    assert seen > 1_800_000_000 and bad == 0, (bad, seen)   # 224 binades x 2^23 + the zeros, +inf and the NaN patterns


def test_decoder_tiny_variances_take_the_slow_sqrt(engine, oracle):
    """Scales so small that a step's auxiliary variance drops below 2^-96: the wave leaves the short square root for sqrtf
    (scaled inputs); still the encoder's sample and the oracle's, bit for bit."""
    n = 600
    mq, sq, mp, sp = oracle.synthetic_latent(31, n)
    tiny = np.float32(1e-16)
    mq, sq, mp, sp = (mq * tiny).astype(np.float32), (sq * tiny).astype(np.float32), (mp * tiny).astype(np.float32), (sp * tiny).astype(np.float32)
    ql, qs, pl, ps = (torch.from_numpy(a[None]).cuda().contiguous() for a in (mq, sq, mp, sp))
    lay = engine.layout(1, n, 300, 42)
    params = engine.params(3.0, 20, 10)
    K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, 64)
    Kh, ih = K.cpu().numpy(), idx.cpu().numpy()
    assert Kh.min() >= 1
    for mode in ("auto", "tables", "fused", "legacy", "tensors"):
        assert torch.equal(engine.decode_blocks(params, lay, pl, ps, 42, K, idx, mode=mode), sample), mode
    blocks = [ih[lay.natural[j], :Kh[lay.natural[j]]].tolist() for j in range(2)]
    assert np.array_equal(oracle.decode_tensor(mp, sp, blocks, 42, 20, block_size=300), sample[0].cpu().numpy())


@pytest.mark.parametrize("n_t,n,bs,omega,eps1,B", [(64, 8192, 1000, 3.0, 1.2, 20), (3, 8192, 1000, 3.0, 1.2, 20),
                                                   (2, 301056, 1000, 3.0, 1.0, 10), (5, 1234, 300, 2.0, 1.0, 1),
                                                   (4, 4099, 4099, 3.0, 1.5, 5), (3, 700, 7, 3.0, 1.0, 10),
                                                   (1, 5000, 1000, 5.0, 1.2, 3), (2, 12288, 1000, 3.0, 1.0, 10),
                                                   # 4100 blocks >= 16 per CU: the legacy decoder's LDS-copy build decode_kernel<true> (round 6:
                                                   # the kernel trace of the suite showed that nothing reached it)
                                                   (4100, 64, 64, 3.0, 1.0, 5)])
def test_decoder_variants_agree_with_the_oracle(engine, oracle, n_t, n, bs, omega, eps1, B):
    """Round 3 decoder (irec_decode.hip): one wave per 256 dims of a block, index path held in a lane vector, rows from the
    per-call proposal tables ("tables") or from the fused Philox draw ("fused"); the round-2 kernel ("legacy") serves calls
    without dim hints.  All three must reproduce the encoder's sample bit for bit, and the oracle's decode_tensor of the same
    indices (beam_search_coder.py:124-148, coder.py:459-491): ragged tails, blocks of more than 1024 dims, 7-dim blocks whose
    rows do not start on a Philox-block boundary, tensors larger than the LDS, S larger than the block count."""
    S = oracle.n_samples(omega, eps1)
    stats = [oracle.synthetic_latent(7000 + i, n) for i in range(n_t)]
    ql, qs, pl, ps = (torch.from_numpy(np.stack([s[k] for s in stats])).cuda().contiguous() for k in range(4))
    lay = engine.layout(n_t, n, bs, 42)
    params = engine.params(omega, S, B)
    max_K = 96
    K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, max_K)
    assert int(K.max()) <= max_K and int(K.min()) >= 0
    modes = ["auto", "tables", "fused", "legacy"]
    if engine.lib.irec_decode_tensors_supported(ctypes.byref(params), n, min(bs, n)):
        modes += ["tensors", "tensors_fused"]
    else:
        assert n > 10000 or lay.n_blocks // n_t > 10      # (tensors beyond the LDS, or of more units than a workgroup takes)
    for mode in modes:
        rec = engine.decode_blocks(params, lay, pl, ps, 42, K, idx, mode=mode)
        assert torch.equal(rec, sample), mode
    # a table window shorter than the longest index path: those blocks draw in the kernel
    short = engine.params(omega, S, B, table_steps=max(1, int(K.max()) // 2))
    assert torch.equal(engine.decode_blocks(short, lay, pl, ps, 42, K, idx), sample)
    Kh, ih = K.cpu().numpy(), idx.cpu().numpy()
    bpt = lay.blocks_per_tensor
    for i in sorted({0, n_t - 1}):
        blocks = [ih[lay.natural[i * bpt + j], :Kh[lay.natural[i * bpt + j]]].tolist() for j in range(bpt)]
        if bpt > 40:                                     # (the oracle decodes a sample of the blocks of a big tensor)
            perm = oracle.tf_shuffle_perm(42, n)
            for j in (0, bpt // 2, bpt - 1):
                lo, hi = oracle.split_blocks(n, bs)[j]
                g = perm[lo:hi]
                want = oracle.decode_block(stats[i][2][g], stats[i][3][g], blocks[j], 42, S)
                assert np.array_equal(sample[i].cpu().numpy()[g], want)
        else:
            want = oracle.decode_tensor(stats[i][2], stats[i][3], blocks, 42, S, block_size=bs)
            assert np.array_equal(sample[i].cpu().numpy(), want)


def test_decoder_leaves_mu_p_on_rows_it_cannot_decode(engine, oracle):
    """A K / index row with K < 0 (the encoder's mark for a block it could not serve) or K > max_K (a block that needs a
    retry) is not decodable: every decode mode returns p.loc on that block's elements -- the staged decoder writes zeros in
    place of the block's variances before the region leaves with mu_p added -- and the exact samples everywhere else."""
    n_t, n, bs = 5, 8192, 1000
    stats = [oracle.synthetic_latent(7100 + i, n) for i in range(n_t)]
    ql, qs, pl, ps = (torch.from_numpy(np.stack([s[k] for s in stats])).cuda().contiguous() for k in range(4))
    lay = engine.layout(n_t, n, bs, 42)
    params = engine.params(3.0, 36, 20)
    K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, 40)
    bpt = lay.blocks_per_tensor
    bad = [(0, 0, -1), (1, 8, 41), (3, 4, -1), (4, 7, 1000)]          # (tensor, block, K written into its row)
    K2 = K.clone()
    for i, j, k in bad:
        K2[int(lay.natural[i * bpt + j])] = k
    want = sample.clone()
    perm = torch.from_numpy(oracle.tf_shuffle_perm(42, n).astype(np.int64)).cuda()
    for i, j, _ in bad:
        lo, hi = oracle.split_blocks(n, bs)[j]
        g = perm[lo:hi]
        want[i, g] = pl[i, g]
    for mode in ("auto", "tensors", "tensors_fused", "tables", "fused", "legacy"):
        rec = engine.decode_blocks(params, lay, pl, ps, 42, K2, idx, mode=mode)
        assert torch.equal(rec, want), mode


def test_skewed_partition_counts_match_the_oracle(engine, oracle):
    """Heavy-tailed K inside one call (per-tensor log-normal scale on delta: K from 1 to 60 and more; real posteriors differ by
    orders of magnitude in KL between residual blocks, resnet_vae.py:462-476): blocks beyond the default table window take the
    second pass, long and short blocks share teams.  Indices, K and samples against the oracle on 12 tensors, as listed and
    with the blocks handed out longest first (Engine.encode_blocks(order_by_K=True)); both orders bit-identical everywhere;
    decode(encode) exact."""
    import bench
    n_t = 96
    q = bench.skewed_batch(n_t, engine.device, 0)
    lay = engine.layout(n_t, 8192, 1000, 42)
    S = 36
    for table_steps in (0, 128):                       # default window (32 steps: the long blocks are deferred) / everything from tables
        params = engine.params(3.0, S, 20, table_steps=table_steps)
        K, idx, sample = engine.encode_blocks(params, lay, *q, 42, 128)
        K2, idx2, sample2 = engine.encode_blocks(params, lay, *q, 42, 128, order_by_K=True)
        Kh = K.cpu().numpy()
        assert Kh.min() >= 0 and Kh.max() >= 40 and Kh.max() <= 128, (int(Kh.min()), int(Kh.max()))
        assert torch.equal(K, K2) and torch.equal(sample, sample2)
        ih, ih2 = idx.cpu().numpy(), idx2.cpu().numpy()
        for r in range(lay.n_blocks):
            assert np.array_equal(ih[r, :Kh[r]], ih2[r, :Kh[r]]), r
        assert torch.equal(engine.decode_blocks(params, lay, q[2], q[3], 42, K, idx), sample)
    c = 12
    hb = [t[:c].cpu().numpy() for t in q]
    ridx, rsamp, _ = oracle.encode_tensors_omp(*hb, 42, 3.0, S, 20, 1000, max_K=128)
    bpt = lay.blocks_per_tensor
    sh = sample[:c].cpu().numpy()
    for i in range(c):
        for j in range(bpt):
            row = lay.natural[i * bpt + j]
            assert ih[row, :Kh[row]].tolist() == ridx[i][j], (i, j)
        assert np.array_equal(sh[i], rsamp[i]), i


def test_decoder_refuses_indices_that_are_no_sample_index(engine, oracle):
    """An index row holding a value outside [0, S) -- a .rec file written with another max_index, a corrupt stream -- must not
    become an out-of-bounds read of the proposal tables (round 3's table path addressed row `index` unchecked): the block is
    not decodable, its elements come out as p.loc in every decode mode, every other block exactly."""
    n_t, n, bs, S = 4, 8192, 1000, 36
    stats = [oracle.synthetic_latent(7300 + i, n) for i in range(n_t)]
    ql, qs, pl, ps = (torch.from_numpy(np.stack([s[k] for s in stats])).cuda().contiguous() for k in range(4))
    lay = engine.layout(n_t, n, bs, 42)
    params = engine.params(3.0, S, 20)
    K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, 40)
    bpt = lay.blocks_per_tensor
    bad = [(0, 0, 0, S), (1, 8, 1, 10 ** 6), (2, 3, 2, -1), (3, 5, 0, 2 ** 31 - 1)]     # (tensor, block, step, value)
    idx2 = idx.clone()
    Kh = K.cpu().numpy()
    want = sample.clone()
    perm = torch.from_numpy(oracle.tf_shuffle_perm(42, n).astype(np.int64)).cuda()
    for i, j, t, v in bad:
        row = int(lay.natural[i * bpt + j])
        idx2[row, min(t, int(Kh[row]) - 1)] = v
        lo, hi = oracle.split_blocks(n, bs)[j]
        g = perm[lo:hi]
        want[i, g] = pl[i, g]
    for mode in ("auto", "tensors", "tensors_fused", "tables", "fused", "legacy"):
        rec = engine.decode_blocks(params, lay, pl, ps, 42, K, idx2, mode=mode)
        assert torch.equal(rec, want), mode
    # values past the block's K are not part of the row: they may hold anything
    idx3 = idx.clone()
    for r in range(lay.n_blocks):
        idx3[r, int(Kh[r]):] = 10 ** 6
    for mode in ("auto", "tables", "fused", "legacy"):
        assert torch.equal(engine.decode_blocks(params, lay, pl, ps, 42, K, idx3, mode=mode), sample), mode


def test_decoder_without_a_dim_bound_leaves_mu_p_on_unlisted_blocks(engine, oracle):
    """irec_beam_decode (no workspace, no max_block_dim) sizes its units from the listed table dims; a block with more dims
    than they cover used to be skipped with its output elements left uninitialised.  It is 'not decodable': p.loc."""
    n = 3000
    mq, sq, mp, sp = oracle.synthetic_latent(7400, n)
    ql, qs, pl, ps = (torch.from_numpy(a[None]).cuda().contiguous() for a in (mq, sq, mp, sp))
    lay = engine.layout(1, n, 1000, 42)
    params = engine.params(3.0, 36, 20)
    K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, 40)
    hinted = engine.params(3.0, 36, 20, table_dims=(200,))           # lists 200-dim blocks only: one 256-dim unit per block
    out = torch.full_like(pl, float("nan"))
    irec_lib = engine.lib
    import irec
    irec._lib.check(irec_lib.irec_beam_decode(engine.ctx, ctypes.byref(hinted), lay.n_blocks, ctypes.c_void_p(lay.block_base.data_ptr()),
                                              ctypes.c_void_p(lay.block_pos.data_ptr()), ctypes.c_void_p(lay.block_dim.data_ptr()),
                                              ctypes.c_void_p(lay.perm.data_ptr()), ctypes.c_void_p(pl.data_ptr()), ctypes.c_void_p(ps.data_ptr()),
                                              42, 40, ctypes.c_void_p(K.data_ptr()), ctypes.c_void_p(idx.data_ptr()),
                                              ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                    "irec_beam_decode")
    assert torch.equal(out, pl)                                        # every block has 1000 dims: none is decodable, none is garbage


@pytest.mark.parametrize("flags", [8, 0, 4, 2], ids=["table", "auto", "one_table", "fused"])
def test_full_size_properties(engine, oracle, flags):
    """BASELINE config 2 at bench size: properties that need no oracle run (round trip, ranges), plus a sampled
    subset of blocks checked against the oracle."""
    n_t, n, bs = 96, 8192, 1000
    stats = [oracle.synthetic_latent(1000 + i, n) for i in range(n_t)]
    ql, qs, pl, ps = (torch.from_numpy(np.stack([s[k] for s in stats])).cuda().contiguous() for k in range(4))
    lay = engine.layout(n_t, n, bs, 42)
    params = engine.params(3.0, 36, 20, flags)
    K, idx, sample = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, 32)
    rec = engine.decode_blocks(params, lay, pl, ps, 42, K, idx)
    assert torch.equal(rec, sample)                                  # decode(encode) == sample, every dim
    Kh, ih = K.cpu().numpy(), idx.cpu().numpy()
    assert Kh.min() >= 1 and Kh.max() <= 32
    for r in range(lay.n_blocks):
        assert (ih[r, :Kh[r]] >= 0).all() and (ih[r, :Kh[r]] < 36).all()
    _, K_kl = engine.block_kl(params, lay, ql, qs, pl, ps)
    assert torch.equal(K_kl, K)
    # second run is bit-identical (no atomics / scheduling dependence in the results)
    K2, idx2, sample2 = engine.encode_blocks(params, lay, ql, qs, pl, ps, 42, 32)
    assert torch.equal(K2, K) and torch.equal(sample2, sample)
    for r in range(lay.n_blocks):
        assert np.array_equal(idx2[r, :Kh[r]].cpu().numpy(), ih[r, :Kh[r]])
    # sampled blocks against the oracle
    perm = oracle.tf_shuffle_perm(42, n)
    rng = np.random.default_rng(0)
    blocks = oracle.split_blocks(n, bs)
    sh = sample.cpu().numpy()
    for t, b in zip(rng.integers(0, n_t, 12), rng.integers(0, len(blocks), 12)):
        lo, hi = blocks[b]
        g = perm[lo:hi]
        ridx, rs = oracle.encode_block(*(s[g] for s in stats[t]), 42, 3.0, 36, 20)
        row = lay.natural[t * lay.blocks_per_tensor + b]
        assert ih[row, :Kh[row]].tolist() == ridx
        assert np.array_equal(sh[t][g], rs)


def _full_size_check(engine, oracle, n_t, n, bs, omega, eps1, B, n_oracle_blocks, gen_seed=5000):
    """encode -> decode round trip on every dim, index ranges, K == ceil(KL / Omega) from the KL kernel, run-to-run
    determinism, and a random subset of blocks against the oracle."""
    S = oracle.n_samples(omega, eps1)
    g = torch.Generator(device="cuda"); g.manual_seed(gen_seed)
    shape = (n_t, n)
    mp = torch.randn(shape, generator=g, device="cuda")
    lsp = 0.25 * torch.randn(shape, generator=g, device="cuda")
    sp = torch.exp(lsp)
    mq = mp + sp * 0.2 * torch.randn(shape, generator=g, device="cuda")
    sq = torch.exp(lsp - (0.05 * torch.randn(shape, generator=g, device="cuda")).abs())
    lay = engine.layout(n_t, n, bs, 42)
    params = engine.params(omega, S, B)
    K, idx, sample = engine.encode_blocks(params, lay, mq, sq, mp, sp, 42, 48)
    Kh = K.cpu().numpy()
    assert Kh.min() >= 1 and Kh.max() <= 48
    assert torch.equal(engine.decode_blocks(params, lay, mp, sp, 42, K, idx), sample)
    _, K_kl = engine.block_kl(params, lay, mq, sq, mp, sp)
    assert torch.equal(K_kl, K)
    ih = idx.cpu().numpy()
    valid = np.arange(ih.shape[1])[None, :] < Kh[:, None]
    assert (ih[valid] >= 0).all() and (ih[valid] < S).all()
    K2, idx2, sample2 = engine.encode_blocks(params, lay, mq, sq, mp, sp, 42, 48)
    assert torch.equal(K2, K) and torch.equal(sample2, sample) and np.array_equal(idx2.cpu().numpy()[valid], ih[valid])
    perm = oracle.tf_shuffle_perm(42, n)
    blocks = oracle.split_blocks(n, bs)
    rng = np.random.default_rng(1)
    host = [t.cpu().numpy() for t in (mq, sq, mp, sp)]
    sh = sample.cpu().numpy()
    for t, b in zip(rng.integers(0, n_t, n_oracle_blocks), rng.integers(0, len(blocks), n_oracle_blocks)):
        lo, hi = blocks[b]
        gsel = perm[lo:hi]
        ridx, rs = oracle.encode_block(*(h[t][gsel] for h in host), 42, omega, S, B)
        row = lay.natural[t * lay.blocks_per_tensor + b]
        assert ih[row, :Kh[row]].tolist() == ridx
        assert np.array_equal(sh[t][gsel], rs)
    # round 3: a run of WHOLE tensors against the OpenMP oracle as well (every block of n_oracle_tensors tensors spread over
    # the batch: VERDICT r2 found ten blocks of 64 800 thin) -- up to 256 tensors, bounded to ~6e10 proposal evaluations of oracle work
    per_tensor = S * n * (1 + max(float(Kh.mean()) - 1, 0) * B)
    n_oracle_tensors = int(max(1, min(n_t, 256, 6e10 // per_tensor)))
    pick = np.unique(np.linspace(0, n_t - 1, n_oracle_tensors).astype(np.int64))
    ridx, rsamp, _ = oracle.encode_tensors_omp(*(h[pick] for h in host), 42, omega, S, B, bs, max_K=48)
    bpt = lay.blocks_per_tensor
    for k, t in enumerate(pick):
        for j in range(bpt):
            row = lay.natural[t * bpt + j]
            assert ih[row, :Kh[row]].tolist() == ridx[k][j], (t, j)
        assert np.array_equal(sh[t], rsamp[k]), t
    return Kh


def test_config3_300_images_full_size(engine, oracle):
    # BASELINE configs[2]: 300 Cifar10-shaped images x 24 latent tensors [16,16,32] each = 7200 latents, 64 800 blocks
    Kh = _full_size_check(engine, oracle, 300 * 24, 8192, 1000, 3.0, 1.2, 20, n_oracle_blocks=10)
    assert len(Kh) == 300 * 24 * 9


def test_config4_kodak_shapes_full_size(engine, oracle):
    # BASELINE configs[3]: Kodak 768 x 512, level 2 [1,8,12,128] = 12 288 dims (13 blocks), level 1 [1,32,48,196] =
    # 301 056 dims (302 blocks); B = 10, Omega = 3, eps = 0 -> S = 20
    Kh2 = _full_size_check(engine, oracle, 1, 12288, 1000, 3.0, 1.0, 10, n_oracle_blocks=4, gen_seed=61)
    Kh1 = _full_size_check(engine, oracle, 1, 301056, 1000, 3.0, 1.0, 10, n_oracle_blocks=6, gen_seed=62)
    assert len(Kh2) == 13 and len(Kh1) == 302


def test_config5_stress_full_size(engine, oracle):
    # BASELINE configs[4]: ImageNet32 RVAE latents, B = 30, Omega = 5 (S = 148): 4440 candidates per step
    _full_size_check(engine, oracle, 96, 8192, 1000, 5.0, 1.0, 30, n_oracle_blocks=6, gen_seed=63)


def test_config5_stress_s403_full_tensors(engine, oracle):
    # BASELINE configs[4] at eps = 0.2: S = int(e^6) = 403, B = 30 -> 12 090 candidates per step (sort keys in the slab, six
    # sample passes): 24 full latent tensors (216 blocks), round trip on every dim + four blocks against the oracle
    Kh = _full_size_check(engine, oracle, 24, 8192, 1000, 5.0, 1.2, 30, n_oracle_blocks=4, gen_seed=64)
    assert len(Kh) == 24 * 9


# ---- round 2: bounded proposal tables, plan export, per-stream scratch ------------------------------------------------
def test_table_window_mixed_partition_counts_fit_256mb(engine, oracle):
    """Blocks with K = 7..8 and blocks with K ~ 3000 in ONE call at S = 403 (Omega = 5, eps = 0.2): the proposal tables
    cover a window bounded in steps and bytes whatever max_K is, the high-K blocks take the fused-Philox second pass, the
    scratch stays under 256 MB and every output is the oracle's, bit for bit."""
    n, bs, omega, eps1, B = 1064, 1000, 5.0, 1.2, 20
    S = oracle.n_samples(omega, eps1)
    assert S == 403
    lat = [list(oracle.synthetic_latent(900 + i, n)) for i in range(3)]
    perm = oracle.tf_shuffle_perm(42, n)
    tail = perm[1000:]                                     # the 64-dim tail block of every tensor
    for i, l in enumerate(lat):
        if i != 1:                                         # tensors 0 and 2: a tail block ~ 235 nats/dim -> K ~ 3000
            l[0][tail] = (l[2][tail] + 21.7 * l[3][tail]).astype(np.float32)
    q = [np.stack([l[j] for l in lat]) for j in range(4)]
    c = _coder(omega, B, eps1, block_size=bs, variant="table")
    lay = engine.layout(3, n, bs, 42)
    plan0 = engine.plan(c._params(), lay, 4096)            # a fresh coder: the default window
    assert plan0["table_steps"] == 32 and plan0["workspace_bytes"] < 256 * 2 ** 20, plan0
    idx, sample = c.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42, batched=True)
    Ks = [[len(b) for b in t] for t in idx]
    assert max(Ks[0]) > 2500 and max(Ks[1]) <= 32 and max(Ks[2]) > 2500, Ks
    for i in range(3):
        ridx, rs = oracle.encode_tensor(q[0][i], q[1][i], q[2][i], q[3][i], 42, omega, S, B, block_size=bs)
        assert idx[i] == ridx, i
        assert np.array_equal(sample[i].cpu().numpy(), rs), i
    # the coder's window hint follows the bulk of the blocks (K = 8), not the two outliers; asked for more than fits, the
    # library bounds the tables' bytes
    assert 8 <= c._params().table_steps <= 16, c._params().table_steps
    c.table_steps = 4000
    plan = engine.plan(c._params(), lay, max(max(k) for k in Ks))
    per_step = 2 * S * (1000 + 64)
    assert plan["table_steps"] == (64 << 20) // per_step and plan["workspace_bytes"] < 256 * 2 ** 20, plan
    idx_b, sample_b = c.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42, batched=True)   # coded with that window
    assert idx_b == idx and torch.equal(sample_b, sample)
    # round trip through the decoder
    rec = c.decode(_normal(q[2], q[3]), idx, seed=42, batched=True)
    assert torch.equal(rec, sample)


@pytest.mark.parametrize("steps", [1, 4, 7])
def test_short_table_window_is_bit_exact(engine, oracle, steps):
    """table_steps below the blocks' K: everything goes through the second pass, or is split between the passes."""
    q = [np.stack([oracle.synthetic_latent(40 + i, 8192)[j] for i in range(8)]) for j in range(4)]
    for variant in ("table", "one_table"):
        c = _coder(3.0, 20, 1.2, block_size=1000, variant=variant)
        c.table_steps = steps
        idx, sample = c.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42, batched=True)
        for i in (0, 5):
            ridx, rs = oracle.encode_tensor(q[0][i], q[1][i], q[2][i], q[3][i], 42, 3.0, 36, 20, block_size=1000)
            assert idx[i] == ridx and np.array_equal(sample[i].cpu().numpy(), rs), (variant, steps, i)


def test_plan_names_the_kernels_that_run(engine, oracle):
    import irec
    S = 36
    big = engine.layout(64, 8192, 1000, 42)
    small = engine.layout(1, 8192, 1000, 42)
    p = engine.params(3.0, S, 20)
    assert engine.plan(p, big, 32)["kernel"] == "encode_team_kernel<20,3,1>"
    assert engine.plan(p, big, 32)["table_kernel"] == "prep_kernel (copy bits)"
    assert engine.plan(p, small, 32)["kernel"].startswith("encode_fast_kernel<20,")       # < 64 blocks: one-table set-up
    assert engine.plan(engine.params(3.0, S, 20, irec._lib.IREC_FLAG_FUSED_PHILOX), big, 32)["table_kernel"] == ""
    # round 6: the reference's default settings (at most ten beams, S * 10 <= 256) run encode_ten_kernel; the
    # team encoder's ten-beam builds keep more samples, shared rows, margins -- and plain calls under IREC_FLAG_NO_TEN
    ten = engine.plan(engine.params(3.0, 20, 10), big, 32)
    assert ten["kernel"] == "encode_ten_kernel<3>" and ten["teams_per_wg"] == 3 and ten["waves_per_wg"] == 12 and ten["lds_bytes"] <= 160 * 1024
    assert engine.plan(engine.params(3.0, 20, 10, irec._lib.IREC_FLAG_NO_TEN), big, 32)["kernel"] == "encode_team_kernel<10,3,1>"
    assert engine.plan(engine.params(3.3, 27, 10), big, 32)["kernel"] == "encode_team_kernel<10,3,1>"      # 270 candidates per step
    assert engine.plan(engine.params(5.0, 148, 30), big, 32)["kernel"] == "encode_team_kernel<30,1,3>"
    assert engine.plan(engine.params(3.0, S, 40), big, 32)["kernel"] == "encode_team_kernel<60,1,3>"     # round 3: 32 < B <= 60
    assert engine.plan(engine.params(3.0, S, 40), small, 32)["kernel"] == "encode_team_kernel<60,1,3>"   # (no one-table encoder there)
    assert engine.plan(engine.params(3.0, S, 50), big, 32)["kernel"] == "encode_team_kernel<60,1,3>"     # B = 50 below S = 128
    assert engine.plan(engine.params(5.0, 148, 50), big, 32)["kernel"] == "encode_team_kernel<54,1,3>"   # round 4: stripes of 18 from S = 128 on
    assert engine.plan(engine.params(5.0, 148, 57), big, 32)["kernel"] == "encode_team_kernel<60,1,3>"
    assert engine.plan(engine.params(3.0, S, 64), big, 32)["kernel"] == "encode_generic_kernel"          # 60 < B <= 64
    assert engine.plan(engine.params(6.0, 8103, 10), big, 32)["kernel"] == "encode_team_kernel<10,3,1,passes>"
    # one beam: one wave per block (irec_lone.hip), 12 blocks in flight per CU; the team encoder's one-beam builds on request
    lone = engine.plan(engine.params(6.0, 8103, 1), big, 32)
    assert lone["kernel"] == "encode_lone_kernel" and lone["waves_per_wg"] == 12 and lone["table_kernel"] == "prep_kernel (copy bits)"
    pin = irec._lib.IREC_FLAG_SHAPE["team"]
    assert engine.plan(engine.params(6.0, 8103, 1, pin), big, 32)["kernel"] == "encode_team_kernel<10,3,1,passes,one>"
    assert engine.plan(engine.params(5.0, 403, 1, pin), big, 32)["kernel"] == "encode_team_kernel<10,3,1,one>"   # 403 samples in one pass
    info = engine.plan(p, big, 32)
    assert info["n_cu"] == 256 and info["clock_mhz"] > 1000 and info["lds_bytes"] <= 160 * 1024


def test_two_streams_encode_concurrently(engine, oracle):
    """One scratch buffer per stream: two encodes in flight at once on different streams do not share block counters,
    tables or slabs."""
    qa = [np.stack([oracle.synthetic_latent(70 + i, 8192)[j] for i in range(16)]) for j in range(4)]
    qb = [np.stack([oracle.synthetic_latent(90 + i, 8192)[j] for i in range(16)]) for j in range(4)]
    c = _coder(3.0, 20, 1.2, block_size=1000, variant="table")
    ta = [torch.as_tensor(a, device="cuda") for a in qa]
    tb = [torch.as_tensor(a, device="cuda") for a in qb]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    pend = []
    for rep in range(3):
        with torch.cuda.stream(s1):
            pa = c.encode_tensors_device(*ta, 42, 1000)
        with torch.cuda.stream(s2):
            pb = c.encode_tensors_device(*tb, 42, 1000)
        pend.append((pa, pb))
    torch.cuda.synchronize()
    assert len(engine._ws) >= 2
    for pa, pb in pend:
        la, lb = pa.to_lists(), pb.to_lists()
        for i in (0, 9):
            ra, rsa = oracle.encode_tensor(qa[0][i], qa[1][i], qa[2][i], qa[3][i], 42, 3.0, 36, 20, block_size=1000)
            rb, rsb = oracle.encode_tensor(qb[0][i], qb[1][i], qb[2][i], qb[3][i], 42, 3.0, 36, 20, block_size=1000)
            assert la[i] == ra and lb[i] == rb
            assert np.array_equal(pa.sample[i].cpu().numpy(), rsa) and np.array_equal(pb.sample[i].cpu().numpy(), rsb)


@pytest.mark.parametrize("shape", ["2", "3", "1x2"])
def test_diagnostic_team_shapes_are_bit_exact(engine, oracle, shape):
    q = [np.stack([oracle.synthetic_latent(20 + i, 8192)[j] for i in range(8)]) for j in range(4)]
    c = _coder(3.0, 20, 1.2, block_size=1000, variant="table")
    c.team_shape = shape
    idx, sample = c.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42, batched=True)
    ridx, rs = oracle.encode_tensor(q[0][3], q[1][3], q[2][3], q[3][3], 42, 3.0, 36, 20, block_size=1000)
    assert idx[3] == ridx and np.array_equal(sample[3].cpu().numpy(), rs)


@pytest.mark.parametrize("B,n_latents,want_W", [(20, 12, 4), (11, 28, 0), (16, 8, 7), (20, 18, 3), (20, 20, 0), (10, 20, 2), (10, 28, 2), (7, 8, 7)])
def test_mid_size_calls_take_the_eight_wave_team(engine, oracle, B, n_latents, want_W):
    """64 .. n_CU blocks: one block per CU at most.  Round 4 (r04x/share_all_probe.log): where the two-team build's 2 n_CU team slots
    hold three partners or more per row (two or more for B <= 10), EVERY row of the call is shared between that many teams
    (irec_encode_plan: the two-team kernel, split = W); otherwise, 10 < B <= 20, the call runs ONE 8-wave beam-striped team per CU.
    A larger call of the same coder takes a multi-team shape without sharing; the oracle's outputs every way."""
    q = [np.stack([oracle.synthetic_latent(300 + i, 8192)[j] for i in range(n_latents)]) for j in range(4)]
    eps1 = 1.2      # (S = 36.  Round 6: with S * 10 <= 256 -- the reference's default S = 20 -- a call of at most ten beams stays whole on
                    #  encode_ten_kernel, whose lone chain is as fast as the shared rows of the team encoder: asserted below)
    S = oracle.n_samples(3.0, eps1)
    c = _coder(3.0, B, eps1, block_size=1000)
    lay = engine.layout(n_latents, 8192, 1000, 42)
    plan = engine.plan(c._params(), lay, 32)
    assert 64 <= lay.n_blocks <= plan["n_cu"], plan
    if B <= 10:
        ten = engine.plan(_coder(3.0, B, 1.0, block_size=1000)._params(), lay, 32)
        assert ten["kernel"] == "encode_ten_kernel<3>" and ten["split"] == 0, ten
    if want_W:
        nb = 10 if B <= 10 else 20
        assert plan["kernel"] == f"encode_team_kernel<{nb},2,1>" and plan["teams_per_wg"] == 2 and plan["split"] == want_W, plan
        assert plan["grid"] * 2 >= lay.n_blocks * want_W and plan["grid"] <= plan["n_cu"], plan
    else:
        assert plan["kernel"] == "encode_team_kernel<20,1,2>" and plan["teams_per_wg"] == 1 and plan["split"] == 0, plan
        assert plan["grid"] == -(-lay.n_blocks // 8) * 8 and plan["waves_per_wg"] == 8   # (a multiple of 8: slot u on XCD u mod 8)
    big = engine.plan(c._params(), engine.layout(64, 8192, 1000, 42), 32)
    assert big["kernel"] != plan["kernel"] and big["teams_per_wg"] >= 2 and big["split"] == 0, big
    idx, sample = c.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42, batched=True)
    assert not c.no_split                                       # (no partner gave up: the call was not coded again without sharing)
    for i in (0, n_latents // 2, n_latents - 1):
        ridx, rs = oracle.encode_tensor(q[0][i], q[1][i], q[2][i], q[3][i], 42, 3.0, S, B, block_size=1000)
        assert idx[i] == ridx and np.array_equal(sample[i].cpu().numpy(), rs), i
    if want_W:   # the same call with every row on one team: the same bits
        c2 = _coder(3.0, B, eps1, block_size=1000)
        c2.no_split = True
        idx2, sample2 = c2.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42, batched=True)
        assert idx2 == idx and torch.equal(sample2, sample)


@pytest.mark.parametrize("B,eps1", [(20, 1.2), (10, 1.0)])
def test_one_to_two_blocks_per_cu_take_the_two_team_shape(engine, oracle, B, eps1):
    """n_CU < blocks <= 2 n_CU with B <= 20 (config 3's per-GPU share: 38 images = 342 blocks per call; the 302 blocks of a
    Kodak image's first level at B = 10): two 4-wave teams per CU at the full register budget instead of three at 168 VGPRs;
    same outputs."""
    n_latents = 38
    S = oracle.n_samples(3.0, eps1)
    q = [np.stack([oracle.synthetic_latent(700 + i, 8192)[j] for i in range(n_latents)]) for j in range(4)]
    c = _coder(3.0, B, eps1, block_size=1000)
    lay = engine.layout(n_latents, 8192, 1000, 42)
    plan = engine.plan(c._params(), lay, 32)
    # (round 6: plain calls of at most ten beams and S * 10 <= 256 run the two-team build of encode_ten_kernel)
    assert plan["n_cu"] < lay.n_blocks <= 2 * plan["n_cu"] and plan["kernel"] == (f"encode_team_kernel<{B},2,1>" if B > 10 else "encode_ten_kernel<2>"), plan
    assert plan["teams_per_wg"] == 2 and plan["grid"] == plan["n_cu"]
    idx, sample = c.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42, batched=True)
    for i in (0, 17, 37):
        ridx, rs = oracle.encode_tensor(q[0][i], q[1][i], q[2][i], q[3][i], 42, 3.0, S, B, block_size=1000)
        assert idx[i] == ridx and np.array_equal(sample[i].cpu().numpy(), rs), i


@pytest.mark.parametrize("B,eps1,n_latents,shape,extra,want_W", [(20, 1.2, 38, "default", 0, 2), (20, 1.2, 38, "3", 0, 5), (20, 1.2, 29, "default", 0, 8),
                                                                (13, 1.2, 45, "3", 0, 3), (10, 1.2, 28, "default", 0, 2),
                                                                (7, 1.2, 14, "default", 0, 4), (20, 1.2, 14, "default", 0, 4)])
def test_calls_of_one_to_two_blocks_per_cu_share_rows_between_teams(engine, oracle, B, eps1, n_latents, shape, extra, want_W):
    """Round 4: in a call of n_CU < blocks < teams * n_CU every CU gets ONE whole block; each row beyond that is coded by W teams
    in the idle team slots of W CUs, which split its samples, exchange their sort keys as tagged granules and then run the same
    selection and update (irec_team.hip, "Shared blocks").  Same outputs as the call with IREC_FLAG_NO_SPLIT (every block on one
    team) bit for bit -- K, index rows, samples of EVERY block --, the oracle's on three tensors; irec_encode_plan reports W;
    a K = 0 tensor and rows beyond the table window among the shared rows; decode(encode) exact."""
    import irec
    S = oracle.n_samples(3.0, eps1)
    stats = [list(oracle.synthetic_latent(900 + i, 8192)) for i in range(n_latents)]
    stats[-1][0], stats[-1][1] = stats[-1][2].copy(), stats[-1][3].copy()        # the last tensor (its rows are shared ones): KL = 0
    ql, qs, pl, ps = (torch.from_numpy(np.stack([s[k] for s in stats])).cuda().contiguous() for k in range(4))
    lay = engine.layout(n_latents, 8192, 1000, 42)
    fl = irec._lib.IREC_FLAG_SHAPE[shape] | extra
    for steps in (0, 6):                               # tables over every partition / a window that the longer rows leave (second pass)
        share = engine.params(3.0, S, B, fl, table_steps=steps)
        whole = engine.params(3.0, S, B, fl | irec._lib.IREC_FLAG_NO_SPLIT, table_steps=steps)
        plan = engine.plan(share, lay, 32)
        assert plan["split"] == want_W and plan["grid"] <= plan["n_cu"] and plan["kernel"].startswith("encode_team_kernel"), plan
        assert engine.plan(whole, lay, 32)["split"] == 0
        K, idx, sample = engine.encode_blocks(share, lay, ql, qs, pl, ps, 42, 32)
        K2, idx2, sample2 = engine.encode_blocks(whole, lay, ql, qs, pl, ps, 42, 32)
        Kh = K.cpu().numpy()
        assert Kh.min() >= 0 and Kh.max() <= 32 and torch.equal(K, K2) and torch.equal(sample, sample2)
        ih, ih2 = idx.cpu().numpy(), idx2.cpu().numpy()
        for r in range(lay.n_blocks):
            assert np.array_equal(ih[r, :Kh[r]], ih2[r, :Kh[r]]), r
        assert torch.equal(engine.decode_blocks(share, lay, pl, ps, 42, K, idx), sample)
    bpt = lay.blocks_per_tensor
    assert all(Kh[lay.natural[(n_latents - 1) * bpt + j]] == 0 for j in range(bpt)) and torch.equal(sample[-1], pl[-1])
    if B <= 10 and not extra:
        return
    for i in (0, n_latents // 2, n_latents - 2):
        ridx, rs = oracle.encode_tensor(*stats[i], 42, 3.0, S, B, block_size=1000)
        assert [ih[lay.natural[i * bpt + j], :Kh[lay.natural[i * bpt + j]]].tolist() for j in range(bpt)] == ridx, i
        assert np.array_equal(sample[i].cpu().numpy(), rs), i


@pytest.mark.parametrize("B,n_latents,shape", [(20, 38, "default"), (20, 45, "default"), (10, 38, "default"), (20, 60, "3"), (20, 29, "default")])
def test_cost_ordered_hand_out_of_mid_size_calls(engine, oracle, B, n_latents, shape):
    """Round 4: a call of more rows than CUs whose rows all find a team at once (one to TEAMS rows per CU) is dealt BY COST: the head
    kernel writes K * dims of every row, a CU's first team takes the row of ascending rank w, the other teams (and the teams sharing a
    row) the costliest rows (irec_team.hip, "Cost-ordered hand-out").  Which CU codes a row never changes what it emits: same K, index
    rows and samples as with IREC_FLAG_LISTED_ORDER, bit for bit, with K from 0 to several dozen inside one call; the oracle's on the
    tensors with the shortest and the longest rows; decode(encode) exact."""
    import irec
    S = oracle.n_samples(3.0, 1.2 if B > 10 else 1.0)
    rng = np.random.default_rng(77)
    stats = []
    for i in range(n_latents):
        mq, sq, mp, sp = oracle.synthetic_latent(2100 + i, 8192)
        f = np.float32(np.exp(np.clip(rng.normal(0.0, 0.6), -1.5, 0.9)))  # per-tensor scale on delta: K differs between tensors
        stats.append(((mp + (mq - mp) * f).astype(np.float32), sq, mp, sp))
    stats[3] = (stats[3][2].copy(), stats[3][3].copy(), stats[3][2], stats[3][3])   # one tensor with KL = 0
    ql, qs, pl, ps = (torch.from_numpy(np.stack([s[k] for s in stats])).cuda().contiguous() for k in range(4))
    lay = engine.layout(n_latents, 8192, 1000, 42)
    fl = irec._lib.IREC_FLAG_SHAPE[shape]
    by_cost = engine.params(3.0, S, B, fl)
    listed = engine.params(3.0, S, B, fl | irec._lib.IREC_FLAG_LISTED_ORDER)
    K, idx, sample = engine.encode_blocks(by_cost, lay, ql, qs, pl, ps, 42, 64)
    K2, idx2, sample2 = engine.encode_blocks(listed, lay, ql, qs, pl, ps, 42, 64)
    Kh = K.cpu().numpy()
    assert Kh.min() == 0 and Kh.max() <= 64 and Kh.max() >= 12, (Kh.min(), Kh.max())
    assert torch.equal(K, K2) and torch.equal(sample, sample2)
    ih, ih2 = idx.cpu().numpy(), idx2.cpu().numpy()
    for r in range(lay.n_blocks):
        assert np.array_equal(ih[r, :Kh[r]], ih2[r, :Kh[r]]), r
    assert torch.equal(engine.decode_blocks(by_cost, lay, pl, ps, 42, K, idx), sample)
    bpt = lay.blocks_per_tensor
    per_tensor = np.array([sum(int(Kh[lay.natural[i * bpt + j]]) for j in range(bpt)) for i in range(n_latents)])
    for i in (int(np.argmax(per_tensor)), int(np.argsort(per_tensor)[1]), 3):
        ridx, rs = oracle.encode_tensor(*stats[i], 42, 3.0, S, B, block_size=1000)
        assert [ih[lay.natural[i * bpt + j], :Kh[lay.natural[i * bpt + j]]].tolist() for j in range(bpt)] == ridx, i
        assert np.array_equal(sample[i].cpu().numpy(), rs), i


def test_shared_rows_give_up_instead_of_hanging(engine, oracle):
    """Test hook (IREC_FLAG_TEST_SPLIT_ORPHAN): the partner teams of every shared row leave at once, so team 0 of each must take the
    give-up exit (100 ms): out_K = -2 on the shared rows, every whole row coded as ever; the Python coder codes the call again
    without sharing (SplitNotResident -> BeamSearchCoder._split_gave_up) and returns the oracle's outputs."""
    import irec
    n_latents, S, B = 29, 36, 20                       # 261 blocks: five shared rows
    stats = [oracle.synthetic_latent(950 + i, 8192) for i in range(n_latents)]
    ql, qs, pl, ps = (torch.from_numpy(np.stack([s[k] for s in stats])).cuda().contiguous() for k in range(4))
    lay = engine.layout(n_latents, 8192, 1000, 42)
    orphan = engine.params(3.0, S, B, irec._lib.IREC_FLAG_TEST_SPLIT_ORPHAN)
    whole = engine.params(3.0, S, B, irec._lib.IREC_FLAG_NO_SPLIT)
    K, idx, sample = engine.encode_blocks(orphan, lay, ql, qs, pl, ps, 42, 32)
    K2, idx2, sample2 = engine.encode_blocks(whole, lay, ql, qs, pl, ps, 42, 32)
    Kh, K2h = K.cpu().numpy(), K2.cpu().numpy()
    assert (Kh == -2).sum() == lay.n_blocks - 256 and (Kh[Kh != -2] == K2h[Kh != -2]).all()
    c = _coder(3.0, B, 1.2, block_size=1000, variant="auto")
    c._test_split_orphan = True
    idx_l, smp = c.encode(_normal(ql, qs), _normal(pl, ps), seed=42, batched=True)
    assert not c.no_split and c._split_strikes == 1 and c._split_pause == 0   # one give-up, one recode without sharing: sharing is back
    for i in (0, n_latents - 1):
        ridx, rs = oracle.encode_tensor(*stats[i], 42, 3.0, S, B, block_size=1000)
        assert idx_l[i] == ridx and np.array_equal(smp[i].cpu().numpy(), rs), i


def test_library_errors_are_coding_errors(engine):
    import irec
    c = irec.BeamSearchCoder(kl_per_partition=3., n_beams=300, extra_samples=1.)     # beyond IREC_MAX_BEAMS = 256
    with pytest.raises(irec.CodingError):
        c.encode(_normal(np.zeros((1, 4), np.float32), np.ones((1, 4), np.float32)),
                 _normal(np.zeros((1, 4), np.float32), np.ones((1, 4), np.float32)), seed=1)
    assert issubclass(irec._lib.IrecLibraryError, irec.CodingError)
    st = irec._lib.load().irec_beam_encode(engine.ctx, None, 1, *([None] * 3), 4, *([None] * 5), 1, 4, *([None] * 4), 0, None)
    with pytest.raises(irec.CodingError):
        irec._lib.check(st, "irec_beam_encode(null params)")


# ---- split encoder: several workgroups share one block of a small call -------------------------------------------------
@pytest.mark.parametrize("n_tensors,n,bs,omega,eps1,B,mode", [(1, 8192, 1000, 3.0, 1.2, 20, "beams"), (1, 1000, None, 3.0, 1.2, 20, "beams"),
                                                              (3, 2048, 1000, 3.0, 1.0, 10, "beams"), (1, 777, None, 2.0, 1.0, 7, "beams"),
                                                              (2, 4096, 1000, 3.0, 1.2, 11, "beams"), (1, 64, None, 1.5, 1.0, 20, "beams"),
                                                              # 33 .. 63 blocks: two or three workgroups per block, more than two beam slots each -- the
                                                              # sample-sharing form (round 6: the flag that forced it on the shapes above is gone)
                                                              (5, 8192, 1000, 3.0, 1.2, 20, "samples"), (4, 8192, 1000, 3.0, 1.0, 10, "samples"),
                                                              (40, 777, None, 2.0, 1.0, 7, "samples")])
def test_split_encoder_small_calls(engine, oracle, n_tensors, n, bs, omega, eps1, B, mode):
    """Calls of few blocks run W workgroups per block (plan['split']) that share the block's beams (each owns at most two
    beam slots: scores every sample for them, forms only its own new beams; plan['split_beams']) or, in the r02b form, its
    samples, and exchange their sort keys every step: the common selection, and therefore every output, is the
    one-workgroup encoder's and the oracle's, bit for bit -- also with K of several hundred steps' worth of cross-workgroup
    hand-offs (in beam mode: of beams read from another workgroup's stores)."""
    S = oracle.n_samples(omega, eps1)
    q = [np.stack([oracle.synthetic_latent(300 + i, n)[j] for i in range(n_tensors)]) for j in range(4)]
    if n == 777 and n_tensors == 1:
        q[1] = (q[1] * 0.3).astype(np.float32)            # sharper posterior: K in the hundreds
    c = _coder(omega, B, eps1, block_size=bs, variant="one_table")
    lay = engine.layout(n_tensors, n, bs, 42)
    plan = engine.plan(c._params(), lay, 32)
    assert plan["split"] >= 2 and plan["grid"] == lay.n_blocks * plan["split"], plan
    assert plan["split_beams"] == (1 if mode == "beams" and -(-B // plan["split"]) <= 2 else 0), plan
    # the split forms are builds of their own (r03n): <NB,4,true,1> shares samples, <NB,4,true,2> beams -- the latter with
    # two table copies in its LDS
    nb = 10 if B <= 10 else 20
    nw = 8 if (plan["split_beams"] and nb == 20) else 4     # (round 4: the 20-beam beam-split build runs 8 waves per workgroup)
    assert plan["kernel"] == f"encode_fast_kernel<{nb},{nw},true,{2 if plan['split_beams'] else 1}>" and plan["waves_per_wg"] == nw, plan
    idx, sample = c.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42, batched=True)
    c2 = _coder(omega, B, eps1, block_size=bs, variant="one_table_nosplit")
    plain = engine.plan(c2._params(), lay, 32)
    assert plain["split"] == 0 and plain["kernel"] == f"encode_fast_kernel<{nb},4,true>", plain
    assert plan["lds_bytes"] == plain["lds_bytes"] + (40024 if plan["split_beams"] else 0), (plan, plain)
    idx2, sample2 = c2.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42, batched=True)
    assert idx == idx2 and torch.equal(sample, sample2)
    for i in range(n_tensors):
        ridx, rs = oracle.encode_tensor(q[0][i], q[1][i], q[2][i], q[3][i], 42, omega, S, B, block_size=bs)
        got = idx[i] if bs is not None else idx[i]
        assert got == ridx and np.array_equal(sample[i].cpu().numpy(), rs), i
    if n == 777 and n_tensors == 1:
        assert len(idx[0]) > 100


def test_split_encoder_gives_up_instead_of_hanging(engine, oracle):
    """The cooperating workgroups of the split encoder wait for each other every step.  If the partners never arrive (test
    hook: they leave at once) the waiting workgroup must reach its exit: after 100 ms it raises the sticky error flag, the
    block comes back with out_K = -2 and the Python layer turns that into a CodingError -- no hung GPU."""
    import time
    import irec
    mq, sq, mp, sp = (torch.as_tensor(a[None], device="cuda") for a in oracle.synthetic_latent(77, 1000))
    lay = engine.layout(1, 1000, None, 42)
    params = engine.params(3.0, 36, 20, irec._lib.IREC_FLAG_ONE_TABLE | 32)      # 32 = IREC_FLAG_TEST_SPLIT_ORPHAN
    assert engine.plan(params, lay, 32)["split"] >= 2
    t0 = time.perf_counter()
    K, idx, sample = engine.encode_blocks(params, lay, mq, sq, mp, sp, 42, 32)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(K.cpu()[0]) == -2 and 0.08 < dt < 2.0, (K.cpu(), dt)            # (COOP_GIVE_UP_TICKS: 100 ms; 2 s until round 3)
    # the same call without the hook works, and the engine is usable afterwards
    c = _coder(3.0, 20, 1.2, variant="one_table")
    i2, s2 = c.encode(_normal(mq, sq), _normal(mp, sp), seed=42)
    ridx, rs = oracle.encode_block(mq.cpu().numpy()[0], sq.cpu().numpy()[0], mp.cpu().numpy()[0], sp.cpu().numpy()[0], 42, 3.0, 36, 20)
    assert [int(v) for v in i2] == ridx and np.array_equal(s2.cpu().numpy()[0], rs)


def test_split_encoder_give_up_is_coded_again_without_the_split(engine, oracle):
    """a split-encoder call whose partner workgroups are not resident reads back K = -2; the coder codes the
    blocks again with one workgroup per block instead of handing the caller a failed image."""
    mq, sq, mp, sp = (torch.as_tensor(a[None], device="cuda") for a in oracle.synthetic_latent(77, 1000))
    c = _coder(3.0, 20, 1.2, variant="one_table")
    c._test_split_orphan = True                              # every split call gives up after its 100 ms wait
    idx, sample = c.encode(_normal(mq, sq), _normal(mp, sp), seed=42)
    assert not c.no_split and c._split_strikes == 1          # the coder stepped back for the recode only (test below: recovery)
    ridx, rs = oracle.encode_block(mq.cpu().numpy()[0], sq.cpu().numpy()[0], mp.cpu().numpy()[0], sp.cpu().numpy()[0], 42, 3.0, 36, 20)
    assert [int(v) for v in idx] == ridx and np.array_equal(sample.cpu().numpy()[0], rs)


def test_recovery_from_a_give_up(engine, oracle):
    """Round 4's review (#7): one give-up used to set no_split for the life of the coder object -- a transient co-tenant cost every
    later small call a factor of two.  Now a bounded back-off (BeamSearchCoder._split_gave_up): the n-th consecutive give-up keeps the
    next 2^(n-1) calls unshared, the first of them being the recode; the call after takes the split encoder AGAIN
    (irec_encode_plan.split > 0), and a shared call that comes back whole resets the count."""
    import irec
    stats = oracle.synthetic_latent(78, 1000)
    mq, sq, mp, sp = (torch.as_tensor(a[None], device="cuda") for a in stats)
    ridx, rs = oracle.encode_block(*stats, 42, 3.0, 36, 20)
    c = _coder(3.0, 20, 1.2, variant="one_table")
    lay = engine.layout(1, 1000, None, 42)

    def plan_split():                                        # what the coder's NEXT call would launch
        return engine.plan(c._params(), lay, 32)["split"]
    assert plan_split() >= 2
    c._test_split_orphan = True                              # a co-tenant: every shared call gives up after its 100 ms
    idx, sample = c.encode(_normal(mq, sq), _normal(mp, sp), seed=42)
    assert [int(v) for v in idx] == ridx and np.array_equal(sample.cpu().numpy()[0], rs)
    assert (c._split_strikes, c._split_pause) == (1, 0) and plan_split() >= 2      # the next call takes the split encoder again
    idx, sample = c.encode(_normal(mq, sq), _normal(mp, sp), seed=42)              # ... and gives up again: two calls unshared,
    assert [int(v) for v in idx] == ridx and (c._split_strikes, c._split_pause) == (2, 1)   # the recode and one more
    assert plan_split() == 0
    c._test_split_orphan = False                             # the co-tenant has left
    idx, sample = c.encode(_normal(mq, sq), _normal(mp, sp), seed=42)              # the pause's last call: unshared
    assert [int(v) for v in idx] == ridx and (c._split_strikes, c._split_pause) == (2, 0) and plan_split() >= 2
    t0 = time.perf_counter()
    idx, sample = c.encode(_normal(mq, sq), _normal(mp, sp), seed=42)              # shared again, comes back whole: count reset
    assert time.perf_counter() - t0 < 0.08                                         # (no 100 ms wait in it)
    assert [int(v) for v in idx] == ridx and np.array_equal(sample.cpu().numpy()[0], rs)
    assert (c._split_strikes, c._split_pause) == (0, 0) and plan_split() >= 2
    # the bound: the pause never exceeds SPLIT_PAUSE_MAX calls; a PERMANENT co-tenant (round 5's advice: a 100 ms stall, a recode and a graph
    # re-capture every 65th call for ever) ends the sharing for good after SPLIT_STRIKES_FINAL give-ups in a row
    for _ in range(c.SPLIT_STRIKES_FINAL - 1):
        c._split_gave_up()
    assert c._split_pause == c.SPLIT_PAUSE_MAX == 64 and not c.no_split
    c._split_gave_up()
    assert c.no_split and plan_split() == 0


def test_workspace_sized_for_the_call(engine, oracle):
    """Round 5's advice: irec_encode_workspace_bytes sizes a call of blocks beyond 1024 dims for every team slot of the device.
    irec_encode_workspace_bytes_for(n_blocks) sizes it for the teams the call launches; irec_beam_encode takes that, the device-wide
    bound, and anything in between -- same K, indices and sample, the oracle's."""
    import ctypes
    from irec import _lib
    stats = oracle.synthetic_latent(8800, 8192)
    ql, qs, pl, ps = (torch.from_numpy(a[None]).cuda().contiguous() for a in stats)
    lay = engine.layout(1, 8192, None, 42)
    params = engine.with_table_dims(engine.params(3.0, 36, 20), lay)
    full = engine.lib.irec_encode_workspace_bytes(engine.ctx, ctypes.byref(params), lay.max_dim, 64)
    mine = engine.lib.irec_encode_workspace_bytes_for(engine.ctx, ctypes.byref(params), lay.n_blocks, lay.max_dim, 64)
    assert 0 < mine < full
    ridx, rs = oracle.encode_tensor(*stats, 42, 3.0, 36, 20, block_size=None)
    out = []
    for size in (mine, (mine + full) // 2 // 256 * 256, full):
        ws = torch.zeros(size, dtype=torch.uint8, device="cuda")
        K = torch.empty(1, dtype=torch.int32, device="cuda"); idx = torch.empty((1, 64), dtype=torch.int32, device="cuda"); sample = torch.empty_like(ql)
        _lib.check(engine.lib.irec_beam_encode(engine.ctx, ctypes.byref(params), 1, lay.block_base.data_ptr(), lay.block_pos.data_ptr(), lay.block_dim.data_ptr(),
                                               lay.max_dim, None, ql.data_ptr(), qs.data_ptr(), pl.data_ptr(), ps.data_ptr(), 42, 64, K.data_ptr(),
                                               idx.data_ptr(), sample.data_ptr(), ws.data_ptr(), ws.numel(), engine._stream()), "irec_beam_encode")
        torch.cuda.synchronize()
        k = int(K[0])
        assert idx[0, :k].tolist() == [int(v) for v in ridx] and np.array_equal(sample.cpu().numpy()[0], rs.reshape(-1)), size
        out.append(k)
    too_small = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")       # (not even the workspace head)
    assert engine.lib.irec_beam_encode(engine.ctx, ctypes.byref(params), 1, lay.block_base.data_ptr(), lay.block_pos.data_ptr(), lay.block_dim.data_ptr(),
                                       lay.max_dim, None, ql.data_ptr(), qs.data_ptr(), pl.data_ptr(), ps.data_ptr(), 42, 64, K.data_ptr(),
                                       idx.data_ptr(), sample.data_ptr(), too_small.data_ptr(), too_small.numel(), engine._stream()) == -4   # IREC_E_WORKSPACE


def test_infinite_kl_block_does_not_poison_the_partition_hint(engine, oracle):
    """a degenerate block (zero prior scale: infinite KL, 10^9 partitions) is a CodingError for ITS tensor and must
    leave the coder's max_K hint usable: the next, ordinary tensor codes as if nothing had happened."""
    import irec
    good = oracle.synthetic_latent(611, 2192)
    bad = [a.copy() for a in good]
    bad[3][:5] = 0.0                                         # p_scale = 0 on a few dims
    c = irec.BeamSearchCoder(kl_per_partition=3., n_beams=20, extra_samples=1.2, block_size=1000)
    with pytest.raises(irec.CodingError):
        c.encode(_normal(*(a[None] for a in bad[:2])), _normal(*(a[None] for a in bad[2:])), seed=42)
    assert c._max_K_hint <= irec._lib.MAX_PARTITIONS
    idx, sample = c.encode(_normal(*(a[None] for a in good[:2])), _normal(*(a[None] for a in good[2:])), seed=42)
    ridx, rs = oracle.encode_tensor(*good, 42, 3.0, 36, 20, block_size=1000)
    assert idx == ridx and np.array_equal(sample.cpu().numpy()[0], rs)


def test_table_session_key_moves_with_every_call(engine, oracle):
    """inside Engine.table_session a call WITHOUT the reuse flag (another coder's settings) lays its slabs over the
    proposal tables; the twin call that follows must rebuild them instead of being told they are present."""
    import irec
    stats = [oracle.synthetic_latent(620 + i, 8192) for i in range(2)]
    q = [torch.from_numpy(np.stack([s[j] for s in stats])).cuda().contiguous() for j in range(4)]
    lay = engine.layout(2, 8192, 1000, 42)
    reuse = engine.params(3.0, 36, 20, irec._lib.IREC_FLAG_REUSE_TABLES)
    other = engine.params(3.0, 36, 20, irec._lib.IREC_FLAG_FUSED_PHILOX)          # no tables: slabs at the head of the scratch
    K0, i0, s0 = engine.encode_blocks(reuse, lay, *q, 42, 32)
    K0, i0, s0 = K0.clone(), i0.clone(), s0.clone()
    with engine.table_session():
        engine.encode_blocks(reuse, lay, *q, 42, 32)
        engine.encode_blocks(reuse, lay, *q, 42, 32)                              # twin: tables present
        engine.encode_blocks(other, lay, *q, 42, 32)                              # writes over the table area
        K1, i1, s1 = engine.encode_blocks(reuse, lay, *q, 42, 32)                 # must NOT skip its table kernels
    assert torch.equal(K1, K0) and torch.equal(s1, s0)
    Kh = K0.cpu().numpy()
    for r in range(lay.n_blocks):
        assert torch.equal(i1[r, :Kh[r]], i0[r, :Kh[r]])


def test_table_window_follows_the_partition_counts(engine, oracle):
    """The Python coder sizes the proposal tables from the K it has read back (32 steps of table for K ~ 8)."""
    import irec
    q = [np.stack([oracle.synthetic_latent(500 + i, 8192)[j] for i in range(8)]) for j in range(4)]
    c = irec.BeamSearchCoder(kl_per_partition=3., n_beams=20, extra_samples=1.2, block_size=1000)
    assert c._params().table_steps == 32                                  # nothing seen yet: the library default
    idx, sample = c.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42, batched=True)
    kmax = max(len(b) for t in idx for b in t)
    assert 8 <= c._params().table_steps <= kmax + 8 and c._params().table_steps < 32
    idx2, sample2 = c.encode(_normal(q[0], q[1]), _normal(q[2], q[3]), seed=42, batched=True)   # coded with the short window
    assert idx2 == idx and torch.equal(sample2, sample)
    ridx, rs = oracle.encode_tensor(q[0][5], q[1][5], q[2][5], q[3][5], 42, 3.0, 36, 20, block_size=1000)
    assert idx2[5] == ridx and np.array_equal(sample2[5].cpu().numpy(), rs)
    # a later batch with much larger K than the window: the second pass codes it, same results as a fresh coder
    q2 = [a.copy() for a in q]
    q2[1] = (q2[1] * 0.35).astype(np.float32)
    idx3, sample3 = c.encode(_normal(q2[0], q2[1]), _normal(q2[2], q2[3]), seed=42, batched=True)
    fresh = irec.BeamSearchCoder(kl_per_partition=3., n_beams=20, extra_samples=1.2, block_size=1000)
    idx4, sample4 = fresh.encode(_normal(q2[0], q2[1]), _normal(q2[2], q2[3]), seed=42, batched=True)
    assert idx3 == idx4 and torch.equal(sample3, sample4) and max(len(b) for t in idx3 for b in t) > 20


def _keep_words(engine):
    """keep[0..3] of the current stream's scratch after the last call (irec_kernels.h, head of the workspace)."""
    ws = engine._ws[int(torch.cuda.current_stream(engine.device).cuda_stream)]
    return ws[:512].view(torch.int32)[8:12].cpu().tolist()


def test_proposal_tables_are_reused_only_when_the_stamp_matches(engine, oracle):
    """IREC_FLAG_REUSE_TABLES: a call whose tables are already in the scratch keeps them (the device compares the stamps);
    another seed, another window, another table kernel or a call without tables in between all force a rebuild -- and every
    result equals the oracle's whichever way the tables came about."""
    q = [np.stack([oracle.synthetic_latent(300 + i, 8192)[j] for i in range(8)]) for j in range(4)]
    t = [torch.as_tensor(a, device="cuda") for a in q]
    ref = {}

    def check(coder, seed, n_t=8):
        idx, sample = coder.encode(_normal(t[0][:n_t], t[1][:n_t]), _normal(t[2][:n_t], t[3][:n_t]), seed=seed, batched=True)
        for i in (0, n_t - 1):
            if (seed, i) not in ref:
                ref[(seed, i)] = oracle.encode_tensor(q[0][i], q[1][i], q[2][i], q[3][i], seed, 3.0, 36, 20, block_size=1000)
            assert idx[i] == ref[(seed, i)][0] and np.array_equal(sample[i].cpu().numpy(), ref[(seed, i)][1]), (seed, i)
        return _keep_words(engine)

    team = _coder(3.0, 20, 1.2, block_size=1000, variant="table")
    team.table_steps = 12
    engine._ws.pop(int(torch.cuda.current_stream(engine.device).cuda_stream), None)   # a fresh (zero-filled) scratch
    assert check(team, 42)[:2] == [0, 0]            # first call: built
    assert check(team, 42)[:2] == [1, 1]            # same key: kept
    assert check(team, 43)[:2] == [0, 0]            # another seed: rebuilt ...
    assert check(team, 42)[:2] == [0, 0]            # ... and the old seed's tables are gone
    assert check(team, 42)[:2] == [1, 1]
    team.table_steps = 16
    assert check(team, 42)[:2] == [0, 0]            # another window (the second table also moves)
    assert check(team, 42)[:2] == [1, 1]
    one = _coder(3.0, 20, 1.2, block_size=1000, variant="one_table")
    one.table_steps = 16
    assert check(one, 42)[:2] == [0, 0]             # same key but the other table kernel (rows without copy bits)
    assert check(one, 42)[:2] == [1, 1]
    assert check(team, 42)[:2] == [0, 0]
    fused = _coder(3.0, 20, 1.2, block_size=1000, variant="fused")
    check(fused, 42)                                 # no tables: its slabs lie over the table area, the stamps are cleared
    assert check(team, 42)[:2] == [0, 0]
    assert check(team, 42)[:2] == [1, 1]
    off = _coder(3.0, 20, 1.2, block_size=1000, variant="table")
    off.table_steps, off.reuse_tables = 16, False
    assert check(off, 42)[:2] == [0, 0]             # without the flag: always rebuilt (and stamped: the next call may keep)
    assert check(team, 42)[:2] == [1, 1]
    assert check(team, 42, n_t=1)[:2] == [1, 1]     # the tables do not depend on the blocks of the call
    auto = _coder(3.0, 20, 1.2, block_size=1000, variant="auto")
    auto.table_steps = 16
    assert check(auto, 42, n_t=1)[:2] == [0, 0]     # 9 blocks without the team flag: the one-table encoder, other rows
    assert check(auto, 42, n_t=1)[:2] == [1, 1]
    assert check(auto, 42)[:2] == [0, 0]            # 72 blocks: the team encoder again


def test_table_reuse_inside_a_replayed_graph(engine, oracle):
    """The stamp comparison runs on the device, so a captured sequence of calls stays correct on every replay -- also after
    other calls have used the same scratch in between."""
    q = [np.stack([oracle.synthetic_latent(330 + i, 8192)[j] for i in range(2)]) for j in range(4)]
    t = [torch.as_tensor(a, device="cuda") for a in q]
    c = _coder(3.0, 20, 1.2, block_size=1000, variant="auto")
    c.table_steps = 12
    refs = [oracle.encode_tensor(q[0][i], q[1][i], q[2][i], q[3][i], 42, 3.0, 36, 20, block_size=1000) for i in range(2)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        c.encode_tensors_device(*t, 42, 1000)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap_stream = int(torch.cuda.current_stream().cuda_stream)
        p1 = c.encode_tensors_device(*t, 42, 1000)
        p2 = c.encode_tensors_device(*t, 42, 1000)      # second call of the capture: keeps the first one's tables
    for rep in range(3):
        g.replay()
        torch.cuda.synchronize()
        keep = engine._ws[cap_stream][:512].view(torch.int32)[8:12].cpu().tolist()
        assert keep[:2] == [1, 1]
        for p in (p1, p2):
            lists = p.to_lists()
            for i in range(2):
                assert lists[i] == refs[i][0] and np.array_equal(p.sample[i].cpu().numpy(), refs[i][1])
        # somebody else codes with another seed on the capture stream's scratch between the replays
        ext = torch.cuda.ExternalStream(cap_stream)
        with torch.cuda.stream(ext):
            c.encode_tensors_device(*t, 7 + rep, 1000)
        torch.cuda.synchronize()


def test_table_session_skips_only_twin_calls(engine, oracle):
    """Engine.table_session(): inside a run of back-to-back calls a twin of the previous call launches no table kernels
    (IREC_FLAG_TABLES_PRESENT); any difference -- seed, window, number of tensors -- makes the call build or check its tables
    as usual.  Every result is the oracle's."""
    q = [np.stack([oracle.synthetic_latent(350 + i, 8192)[j] for i in range(2)]) for j in range(4)]
    t = [torch.as_tensor(a, device="cuda") for a in q]
    refs = {seed: [oracle.encode_tensor(q[0][i], q[1][i], q[2][i], q[3][i], seed, 3.0, 36, 20, block_size=1000) for i in range(2)]
            for seed in (42, 43)}
    c = _coder(3.0, 20, 1.2, block_size=1000, variant="auto")
    c.table_steps = 12
    with engine.table_session():
        for seed, n_t in ((42, 2), (42, 2), (42, 2), (43, 2), (42, 2), (42, 1), (42, 1), (42, 2)):
            idx, sample = c.encode(_normal(t[0][:n_t], t[1][:n_t]), _normal(t[2][:n_t], t[3][:n_t]), seed=seed, batched=True)
            for i in range(n_t):
                assert idx[i] == refs[seed][i][0] and np.array_equal(sample[i].cpu().numpy(), refs[seed][i][1]), (seed, n_t, i)
    assert getattr(engine._tls, "session", None) is None


def test_fitted_auxiliary_ratios_as_data(engine, oracle):
    """Round 4's review (missing #4): a coder built with extrapolate_auxiliary_ratios=False reads its ratios from a variable a checkpoint
    restores (coder.py:203-231).  The fitter stays out of scope; the fitted ratios are data: BeamSearchCoder.set_auxiliary_variance_ratios
    -> irec_create_with(aux_ratios).  Same error behaviour as the reference: not initialised, and KL beyond the table."""
    import irec
    from irec.coding import CodingError
    c = irec.BeamSearchCoder(kl_per_partition=3., n_beams=20, extra_samples=1.2, extrapolate_auxiliary_ratios=False, block_size=1000)
    stats = oracle.synthetic_latent(5150, 8192)
    q, p = _normal(stats[0][None], stats[1][None]), _normal(stats[2][None], stats[3][None])
    with pytest.raises(CodingError, match="has not been initialized yet"):
        c.encode(q, p, seed=42)
    ratios = (0.9 * (np.arange(12) + 1.0) ** -0.7).astype(np.float32)      # some fitted table of 12 entries: not the power law
    ratios[0] = 1.0
    c.set_auxiliary_variance_ratios(ratios)
    assert c.get_auxiliary_ratio(3) == ratios[3]
    with pytest.raises(CodingError, match="higher than auxiliary variables can account for"):
        c.get_auxiliary_ratio(12)
    idx, sample = c.encode(q, p, seed=42)
    oracle.set_aux_ratios(ratios)
    try:
        ridx, rs = oracle.encode_tensor(*stats, 42, 3.0, 36, 20, block_size=1000)
        assert idx == ridx and np.array_equal(sample.cpu().numpy()[0], rs)
        oracle.set_aux_ratios(None)
        pidx, _ = oracle.encode_tensor(*stats, 42, 3.0, 36, 20, block_size=1000)
        assert pidx != ridx                                                   # (the table matters)
    finally:
        oracle.set_aux_ratios(None)
    assert torch.equal(c.decode(p, idx, seed=42), sample)
    assert c._engine_for(q.loc).max_partitions == 12
    # decoding an index list longer than the fitted table: the reference's error (beam_search_coder.py:129-131 asks for ratio len - 1), not a
    # silent p.loc (round 5's advice)
    too_long = [list(b) for b in idx]
    too_long[0] = list(too_long[0]) + [0] * (13 - len(too_long[0]))
    with pytest.raises(CodingError, match="Maximum possible number of partitions is 12"):
        c.decode(p, too_long, seed=42)
    # the table as an attribute (a checkpoint restore): the device context follows it
    ratios2 = ratios.copy(); ratios2[1:] *= 0.97
    c.aux_variable_variance_ratios = ratios2
    idx2, sample2 = c.encode(q, p, seed=42)
    oracle.set_aux_ratios(ratios2)
    try:
        ridx2, rs2 = oracle.encode_tensor(*stats, 42, 3.0, 36, 20, block_size=1000)
    finally:
        oracle.set_aux_ratios(None)
    assert idx2 == ridx2 and np.array_equal(sample2.cpu().numpy()[0], rs2)
    c.aux_variable_variance_ratios = ratios
    # a block whose KL asks for more partitions than the table has: the reference's error, nothing coded
    sharp = oracle.synthetic_latent(5151, 8192)
    sq = (sharp[1] * 0.25).astype(np.float32)                                # KL ~ 1.4 nats per dim more: K far beyond 12
    with pytest.raises(CodingError, match="Maximum possible number of partitions is 12"):
        c.encode(_normal(sharp[0][None], sq[None]), _normal(sharp[2][None], sharp[3][None]), seed=42)
    # the default engine of the device is untouched
    d = _coder(3.0, 20, 1.2, block_size=1000, variant="auto")
    didx, _ = d.encode(q, p, seed=42)
    assert didx == pidx
