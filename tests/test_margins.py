"""Top-B margins (round 5): irec_beam_encode_ex reports, per block, how close its selections were -- the encoder-side exposure of
index parity to another float32 summation order (include/irec.h; beam_search_coder.py:85-89,118-122).

CPU: the oracle's C margin equals a numpy restatement over its traced scores.
GPU (-m gpu): the device's four floats equal the oracle's bit for bit on the golden blocks and on every kernel that serves the flag
(the margin builds of the team encoder, the generic kernel, the generic second pass), and the flag changes no index, K or sample."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_files


def test_oracle_margins_equal_the_traced_scores(oracle):
    for k, (om, e1, B, D) in enumerate([(3.0, 1.2, 20, 1000), (3.0, 1.0, 10, 192), (5.0, 1.0, 30, 700), (3.0, 1.2, 1, 500),
                                        (2.0, 1.0, 50, 300), (3.0, 1.2, 20, 30), (6.0, 1.0, 10, 257)]):
        S = oracle.n_samples(om, e1)
        st = oracle.synthetic_latent(900 + k, D)
        idx, smp, tr, mg = oracle.encode_block(*st, 42, om, S, B, trace=True, margins=True)
        assert np.array_equal(mg, oracle.margins_from_trace(tr, S, B)), (om, e1, B, D)
        idx2, smp2 = oracle.encode_block(*st, 42, om, S, B)
        assert idx == idx2 and np.array_equal(smp, smp2)
        assert mg[0] >= 0 and mg[2] >= 0 and (len(idx) > 1 or np.isinf(mg[0]))


def _dev(stats):
    return tuple(torch.from_numpy(np.ascontiguousarray(np.stack([s[k] for s in stats]))).cuda().contiguous() for k in range(4))


@pytest.mark.gpu
@pytest.mark.parametrize("path", golden_files("block"), ids=os.path.basename)
def test_golden_block_margins(engine, oracle, path):
    g = np.load(path)
    omega, B, S, seed = float(g["kl_per_partition"]), int(g["n_beams"]), int(g["n_samples"]), int(g["seed"])
    stats = tuple(np.ascontiguousarray(g[k], dtype=np.float32) for k in ("q_loc", "q_scale", "p_loc", "p_scale"))
    n = stats[0].size
    ridx, rs, rm = oracle.encode_block(*stats, seed, omega, S, B, margins=True)
    assert ridx == g["indices"].tolist()
    lay = engine.layout(1, n, None, seed)
    q = tuple(torch.from_numpy(a[None]).cuda().contiguous() for a in stats)
    max_K = max(32, len(ridx))
    K, idx, sample, mg = engine.encode_blocks_margins(engine.params(omega, S, B), lay, *q, seed, max_K)
    assert int(K.cpu()[0]) == len(ridx) and idx.cpu().numpy()[0, :len(ridx)].tolist() == ridx
    assert np.array_equal(sample.cpu().numpy()[0], rs)
    assert np.array_equal(mg.cpu().numpy()[0], rm), (mg.cpu().numpy()[0], rm)


@pytest.mark.gpu
@pytest.mark.parametrize("omega,eps1,B,n_lat,kernel", [
    (3.0, 1.2, 20, 16, "encode_team_kernel<20,3,1,margins>"),      # the BASELINE workload's build (144 blocks)
    (3.0, 1.2, 20, 38, "encode_team_kernel<20,2,1,margins>"),      # one GPU's share of config 3 (342 blocks: rows dealt by cost)
    (3.0, 1.2, 20, 1, "encode_team_kernel<20,3,1,margins>"),       # one image's residual block: no split encoder under the flag
    (3.0, 1.0, 10, 12, "encode_team_kernel<10,3,1,margins>"),      # config 4's settings
    (5.0, 1.0, 30, 3, "encode_team_kernel<30,1,3,margins>"),       # config 5 (S = 148)
    (3.0, 1.2, 1, 3, "encode_generic_kernel (margins)"),           # one beam: the winner's lead at every step
    (2.0, 1.0, 50, 2, "encode_generic_kernel (margins)"),          # S = 7 < B
    (3.0, 1.2, 13, 4, "encode_team_kernel<20,3,1,margins>"),       # fewer beams than the build holds
])
def test_margins_of_batched_calls(engine, oracle, omega, eps1, B, n_lat, kernel):
    S = oracle.n_samples(omega, eps1)
    stats = [oracle.synthetic_latent(4100 + i, 8192) for i in range(n_lat)]
    q = _dev(stats)
    lay = engine.layout(n_lat, 8192, 1000, 42)
    params = engine.params(omega, S, B)
    assert engine.plan(params, lay, 32, margins=True)["kernel"] == kernel
    K, idx, sample, mg = engine.encode_blocks_margins(params, lay, *q, 42, 32)
    K0, idx0, sample0 = engine.encode_blocks(params, lay, *q, 42, 32)
    Kh, ih, mh = K.cpu().numpy(), idx.cpu().numpy(), mg.cpu().numpy()
    assert np.array_equal(Kh, K0.cpu().numpy()) and torch.equal(sample, sample0)            # the flag changes nothing that is emitted
    for r in range(lay.n_blocks):
        assert np.array_equal(ih[r, :Kh[r]], idx0.cpu().numpy()[r, :Kh[r]])
    ridx, rsmp, used, rm = oracle.encode_tensors_omp(*(np.stack([s[k] for s in stats]) for k in range(4)), 42, omega, S, B, 1000,
                                                     margins=True)
    bpt = lay.blocks_per_tensor
    for i in range(n_lat):
        for j in range(bpt):
            r = lay.natural[i * bpt + j]
            assert ih[r, :Kh[r]].tolist() == ridx[i][j], (i, j)
            assert np.array_equal(mh[r], rm[i, j]), (i, j, mh[r], rm[i, j])
    assert np.array_equal(sample.cpu().numpy(), rsmp)
    finite = mh[:, 0][np.isfinite(mh[:, 0])]
    assert finite.size and (finite >= 0).all() and (mh[:, 2] >= 0).all()


@pytest.mark.gpu
def test_margins_of_blocks_the_generic_kernel_codes(engine, oracle):
    """Blocks of more than 1024 dims, and blocks beyond the table window (the second pass of a margins call is the generic kernel's)."""
    stats = [oracle.synthetic_latent(4300 + i, 3000) for i in range(3)]
    q = _dev(stats)
    lay = engine.layout(3, 3000, None, 7)
    params = engine.params(3.0, 20, 10)
    assert engine.plan(params, lay, 64, margins=True)["kernel"] == "encode_generic_kernel (margins)"
    K, idx, sample, mg = engine.encode_blocks_margins(params, lay, *q, 7, 64)
    for i in range(3):
        ridx, rs, rm = oracle.encode_block(*stats[i], 7, 3.0, 20, 10, margins=True)
        r = lay.natural[i]
        assert idx.cpu().numpy()[r, :len(ridx)].tolist() == ridx and np.array_equal(mg.cpu().numpy()[r], rm)
    # a table window of three steps: the long blocks go to the second pass, every block reports its margins
    stats = [oracle.synthetic_latent(4400 + i, 2192) for i in range(70)]
    q = _dev(stats)
    lay = engine.layout(70, 2192, 1000, 42)
    params = engine.params(3.0, 36, 20, table_steps=3)
    plan = engine.plan(params, lay, 32, margins=True)
    assert plan["kernel"] == "encode_team_kernel<20,3,1,margins>" and plan["table_steps"] == 3
    K, idx, sample, mg = engine.encode_blocks_margins(params, lay, *q, 42, 32)
    Kh, mh = K.cpu().numpy(), mg.cpu().numpy()
    assert (Kh > 3).any() and (Kh <= 3).any()
    ridx, rsmp, used, rm = oracle.encode_tensors_omp(*(np.stack([s[k] for s in stats]) for k in range(4)), 42, 3.0, 36, 20, 1000,
                                                     margins=True)
    for i in range(70):
        for j in range(3):
            r = lay.natural[i * 3 + j]
            assert idx.cpu().numpy()[r, :Kh[r]].tolist() == ridx[i][j] and np.array_equal(mh[r], rm[i, j]), (i, j)


@pytest.mark.gpu
def test_margins_flag_and_pointer_go_together(engine):
    import ctypes
    import irec
    lib = irec._lib.load()
    lay = engine.layout(1, 64, None, 1)
    x = torch.ones(64, device="cuda")
    K = torch.zeros(1, dtype=torch.int32, device="cuda")
    idx = torch.zeros((1, 8), dtype=torch.int32, device="cuda")
    mg = torch.zeros((1, 4), device="cuda")
    ws = torch.zeros(1 << 24, dtype=torch.uint8, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    for flags, margin in ((0, mg), (irec._lib.IREC_FLAG_MARGINS, None)):
        params = engine.with_table_dims(engine.params(3.0, 36, 20, flags), lay)
        st = lib.irec_beam_encode_ex(engine.ctx, ctypes.byref(params), 1, P(lay.block_base), P(lay.block_pos), P(lay.block_dim), 64, None,
                                     P(x), P(x), P(x), P(x), 1, 8, P(K), P(idx), P(x.clone()), P(margin) if margin is not None else None,
                                     P(ws), ws.numel(), None)
        assert st == irec._lib.IREC_E_INVALID and b"go together" in lib.irec_last_error()
    params = engine.with_table_dims(engine.params(3.0, 36, 20, irec._lib.IREC_FLAG_MARGINS), lay)
    st = lib.irec_beam_encode(engine.ctx, ctypes.byref(params), 1, P(lay.block_base), P(lay.block_pos), P(lay.block_dim), 64, None,
                              P(x), P(x), P(x), P(x), 1, 8, P(K), P(idx), P(x.clone()), P(ws), ws.numel(), None)
    assert st == irec._lib.IREC_E_INVALID
