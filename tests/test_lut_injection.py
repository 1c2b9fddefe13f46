"""The quantile table is injectable at the boundary (irec_create_ex, include/irec.h; oracle twin: irec_oracle_set_lut).

The 10006 values `dist.quantile` takes at beam_search_coder.py:48-49 are the one TF-dependent primitive of the hot path
this repository can only restate (TFP 0.9's float32 ndtri, evaluated by TensorFlow over Eigen's vectorised log; here over a
correctly rounded one).  A maintainer with a TF 2.1 machine dumps the table (scripts/make_tf_vectors.py: `quantile` of
tf_primitives.npz) and creates the context with it.  These tests prove the hook is live on both sides: a table perturbed
by one ulp in 50 tail entries, injected into library AND oracle, keeps them bit-identical to each other and moves the
outputs away from the default-table run."""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, golden_files


def perturbed_lut(oracle):
    """The restated table with 50 TAIL entries (k = 1..25 and 9982..10006: |z| > 2.8, the branch of ndtri that goes through
    the log) moved by one ulp, alternately up and down."""
    lut = oracle.build_lut().copy()
    ks = list(range(1, 26)) + list(range(10007 - 25, 10007))
    for n, k in enumerate(ks):
        lut[k] = np.nextafter(lut[k], np.float32(np.inf if n % 2 == 0 else -np.inf), dtype=np.float32)
    assert int((lut != oracle.build_lut()).sum()) == 50
    return lut


def _block_fixtures():
    out = []
    for f in golden_files("block"):
        g = np.load(f)
        if int(g["q_loc"].size) >= 192:        # (a one-dim block meets a tail entry once in a hundred runs)
            out.append(f)
    assert len(out) >= 6
    return out


def test_oracle_hook_is_live_and_restores(oracle):
    lut = perturbed_lut(oracle)
    moved = 0
    try:
        for f in _block_fixtures():
            g = np.load(f)
            args = (g["q_loc"], g["q_scale"], g["p_loc"], g["p_scale"], int(g["seed"]), float(g["kl_per_partition"]),
                    int(g["n_samples"]), int(g["n_beams"]))
            oracle.set_lut(None)
            idx0, smp0 = oracle.encode_block(*args)
            assert idx0 == g["indices"].tolist() and np.array_equal(smp0, g["sample"])        # the default table: the fixture
            oracle.set_lut(lut)
            idx1, smp1 = oracle.encode_block(*args)
            dec1 = oracle.decode_block(g["p_loc"], g["p_scale"], idx1, int(g["seed"]), int(g["n_samples"]))
            assert np.array_equal(dec1, smp1)                                                   # round trip under the injected table
            moved += int(idx1 != idx0 or not np.array_equal(smp1, smp0))
    finally:
        oracle.set_lut(None)
    assert moved >= 1, "a table perturbed in 50 entries changed nothing: the oracle does not read the injected table"
    g = np.load(_block_fixtures()[0])
    idx, smp = oracle.encode_block(g["q_loc"], g["q_scale"], g["p_loc"], g["p_scale"], int(g["seed"]), float(g["kl_per_partition"]),
                                   int(g["n_samples"]), int(g["n_beams"]))
    assert idx == g["indices"].tolist() and np.array_equal(smp, g["sample"])                    # restored


def test_create_ex_rejects_tables_that_are_not_numbers():
    """Host-side argument check: runs without a GPU (the table is validated before the device is looked for)."""
    import irec
    lib = irec._lib.load()
    bad = np.zeros(10007, np.float32)
    bad[5000] = np.nan
    ctx = ctypes.c_void_p()
    st = lib.irec_create_ex(0, bad.ctypes.data_as(ctypes.c_void_p), ctypes.byref(ctx))
    assert st == irec._lib.IREC_E_INVALID and not ctx.value
    assert b"lut10007[5000]" in lib.irec_last_error()


@pytest.mark.gpu
def test_injected_table_reaches_every_kernel(engine, oracle):
    """Perturbed table into library (Engine(lut=...)) and oracle: indices, K and samples stay bit-identical between the two on
    every block fixture (team, one-table, split, fused and generic encoders; tensor and block decoders) and on an RVAE-shape
    tensor, and differ from the default-table run on at least one fixture."""
    import torch
    import irec
    lut = perturbed_lut(oracle)
    with pytest.raises(ValueError):
        irec.Engine(engine.device, lut=lut[:100])          # not a table of 10007 entries
    eng = irec.Engine(engine.device, lut=lut)
    moved = 0
    try:
        oracle.set_lut(lut)
        for f in _block_fixtures():
            g = np.load(f)
            omega, S, B = float(g["kl_per_partition"]), int(g["n_samples"]), int(g["n_beams"])
            want_idx, want_smp = oracle.encode_block(g["q_loc"], g["q_scale"], g["p_loc"], g["p_scale"], int(g["seed"]), omega, S, B)
            moved += int(want_idx != g["indices"].tolist() or not np.array_equal(want_smp, g["sample"]))
            q = torch.distributions.Normal(torch.as_tensor(g["q_loc"][None]).cuda(), torch.as_tensor(g["q_scale"][None]).cuda(), validate_args=False)
            p = torch.distributions.Normal(torch.as_tensor(g["p_loc"][None]).cuda(), torch.as_tensor(g["p_scale"][None]).cuda(), validate_args=False)
            for variant in ("table", "auto", "one_table_nosplit", "fused", "generic"):
                if B > 32 and variant in ("one_table_nosplit", "fused"):
                    continue
                c = irec.BeamSearchCoder(kl_per_partition=omega, n_beams=B, extra_samples=float(g["extra_samples"]), engine=eng)
                c.team = variant == "table"
                c.one_table = c.no_split = variant == "one_table_nosplit"
                c.fused_philox = variant == "fused"
                c.force_generic = variant == "generic"
                idx, smp = c.encode(q, p, seed=int(g["seed"]))
                assert [int(i) for i in idx] == want_idx, (os.path.basename(f), variant)
                assert np.array_equal(smp.cpu().numpy().reshape(-1), want_smp), (os.path.basename(f), variant)
                dec = c.decode(p, [int(i) for i in idx], seed=int(g["seed"]))
                assert torch.equal(dec, smp), (os.path.basename(f), variant)
        # a whole tensor through split / merge and the tensor-staged decoder
        mq, sq, mp, sp = oracle.synthetic_latent(3, 8192)
        want_idx, want_smp = oracle.encode_tensor(mq, sq, mp, sp, 42, 3., 36, 20, block_size=1000)
        c = irec.BeamSearchCoder(kl_per_partition=3., n_beams=20, extra_samples=1.2, block_size=1000, engine=eng)
        q = torch.distributions.Normal(torch.from_numpy(mq[None]).cuda(), torch.from_numpy(sq[None]).cuda(), validate_args=False)
        p = torch.distributions.Normal(torch.from_numpy(mp[None]).cuda(), torch.from_numpy(sp[None]).cuda(), validate_args=False)
        idx, smp = c.encode(q, p, seed=42)
        assert idx == want_idx and np.array_equal(smp.cpu().numpy().reshape(-1), want_smp)
        assert torch.equal(c.decode(p, idx, seed=42), smp)
        # the default engine is untouched by the private one
        d = irec.BeamSearchCoder(kl_per_partition=3., n_beams=20, extra_samples=1.2, block_size=1000)
        oracle.set_lut(None)
        ref_idx, ref_smp = oracle.encode_tensor(mq, sq, mp, sp, 42, 3., 36, 20, block_size=1000)
        idx0, smp0 = d.encode(q, p, seed=42)
        assert idx0 == ref_idx and np.array_equal(smp0.cpu().numpy().reshape(-1), ref_smp)
        moved += int(idx0 != idx or not torch.equal(smp0, smp))
    finally:
        oracle.set_lut(None)
    assert moved >= 1, "the injected table changed no output: the hook is dead"
