"""The oracle and the HIP path against outputs of the REFERENCE'S OWN PYTHON.

tests/golden/refpy_reference_outputs.npz was produced by tests/golden/make_golden_refpy.py: /root/reference/rec/coding imported
unmodified and run on the committed fixtures, its TensorFlow / TFP calls served by the numpy stubs of oracle/tfshim (Philox,
shuffle, normal streams and TFP's float32 ndtri from the C oracle).  These tests pin the oracle's (and through it the
kernels') reading of every line of the reference's Python -- hashing, `seed + iteration`, floormod, the flat candidate index,
% and //, gather_nd, index-path bookkeeping, split / merge, the decoder -- to the reference itself.  They do not pin the
TensorFlow primitives: see tests/test_tf_vectors.py for that.

In the build container (where /root/reference exists) the reference is also run LIVE on fresh random inputs."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = np.load(os.path.join(GOLDEN_DIR, "refpy_reference_outputs.npz"))
BLOCKS = [str(n) for n in REF["block_names"]]


def _fixture(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


@pytest.mark.both_suites
@pytest.mark.usefixtures("suite")
@pytest.mark.parametrize("name", BLOCKS)
def test_oracle_equals_reference_python_on_block(oracle, name):
    f = _fixture(name)
    S, B = int(f["n_samples"]), int(f["n_beams"])
    args = (f["q_loc"], f["q_scale"], f["p_loc"], f["p_scale"], int(f["seed"]), float(f["kl_per_partition"]), S, B)
    want = REF[f"{name}_indices"].tolist()
    for mode in (oracle.LITERAL, oracle.CANONICAL):
        idx, sample = oracle.encode_block(*args, mode=mode)
        assert idx == want, (name, mode)
        assert np.allclose(sample, REF[f"{name}_sample"], rtol=0, atol=1e-5)            # north_star: 1e-5 on reconstructions
    assert np.allclose(oracle.decode_block(f["p_loc"], f["p_scale"], want, int(f["seed"]), S), REF[f"{name}_decoded"],
                       rtol=0, atol=1e-5)
    assert np.allclose(REF[f"{name}_decoded"], REF[f"{name}_sample"], rtol=0, atol=1e-5)  # the reference's own round trip
    assert float(REF[f"{name}_codelength"]) == pytest.approx(oracle.codelength(want, S), rel=1e-12)


@pytest.mark.both_suites
@pytest.mark.usefixtures("suite")
def test_oracle_equals_reference_python_on_split_tensor(oracle):
    """GaussianCoder.encode with block_size: the reference's split (tf.random.shuffle) -> 9 encode_block calls -> merge."""
    f = _fixture("tensor_rvae_cfg2")
    idx, sample = oracle.encode_tensor(f["q_loc"], f["q_scale"], f["p_loc"], f["p_scale"], int(f["seed"]),
                                       float(f["kl_per_partition"]), int(f["n_samples"]), int(f["n_beams"]),
                                       block_size=int(f["block_size"]))
    K = REF["tensor_K"]
    assert [len(i) for i in idx] == K.tolist()
    for r, ix in enumerate(idx):
        assert ix == REF["tensor_indices"][r, :K[r]].tolist(), r
    assert np.allclose(sample, REF["tensor_sample"], rtol=0, atol=1e-5)
    assert np.allclose(REF["tensor_decoded"], REF["tensor_sample"], rtol=0, atol=1e-5)


@pytest.mark.both_suites
@pytest.mark.usefixtures("suite")
def test_importance_sampler_equals_reference_python(oracle):
    """encode / decode_gaussian_importance_sample (importance_sampling.py:9-103), alpha = inf and the Gumbel-max branch."""
    import torch
    from irec.coding.samplers import ImportanceSampler
    for c in range(int(REF["n_importance_cases"])):
        t_loc, t_scale, p_loc, p_scale = REF[f"is{c}_in"]
        bits, seed, alpha = REF[f"is{c}_meta"]
        want, want_sample = int(REF[f"is{c}_index"]), REF[f"is{c}_sample"]
        ridx, rsample = oracle.importance_encode(t_loc, t_scale, p_loc, p_scale, float(bits), int(seed), alpha=float(alpha))
        assert ridx == want and np.allclose(rsample, want_sample, rtol=0, atol=1e-5), c
        s = ImportanceSampler(coding_bits=float(bits), alpha=float(alpha))
        N = lambda a, b: torch.distributions.Normal(torch.from_numpy(a.copy()), torch.from_numpy(b.copy()), validate_args=False)  # noqa: E731
        idx, sample = s.coded_sample(N(t_loc, t_scale), N(p_loc, p_scale), int(seed))      # the product's C++ entry point
        assert idx == want and np.allclose(sample.numpy(), want_sample, rtol=0, atol=1e-5), c
        assert np.allclose(s.decode_sample(N(p_loc, p_scale), idx, int(seed)).numpy(), REF[f"is{c}_decoded"], rtol=0, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", BLOCKS)
def test_hip_equals_reference_python_on_block(engine, name):
    import irec
    import torch
    f = _fixture(name)
    c = irec.BeamSearchCoder(kl_per_partition=float(f["kl_per_partition"]), n_beams=int(f["n_beams"]),
                             extra_samples=float(f["extra_samples"]))
    N = lambda a, b: torch.distributions.Normal(torch.as_tensor(a[None]).cuda(), torch.as_tensor(b[None]).cuda(), validate_args=False)  # noqa: E731
    idx, sample = c.encode(N(f["q_loc"], f["q_scale"]), N(f["p_loc"], f["p_scale"]), seed=int(f["seed"]))
    assert [int(i) for i in idx] == REF[f"{name}_indices"].tolist()
    assert np.allclose(sample.cpu().numpy().reshape(-1), REF[f"{name}_sample"], rtol=0, atol=1e-5)
    dec = c.decode(N(f["p_loc"], f["p_scale"]), [int(i) for i in idx], seed=int(f["seed"]))
    assert np.allclose(dec.cpu().numpy().reshape(-1), REF[f"{name}_decoded"], rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_hip_equals_reference_python_on_split_tensor(engine):
    import irec
    import torch
    f = _fixture("tensor_rvae_cfg2")
    c = irec.BeamSearchCoder(kl_per_partition=float(f["kl_per_partition"]), n_beams=int(f["n_beams"]),
                             extra_samples=float(f["extra_samples"]), block_size=int(f["block_size"]))
    N = lambda a, b: torch.distributions.Normal(torch.as_tensor(a).cuda(), torch.as_tensor(b).cuda(), validate_args=False)  # noqa: E731
    idx, sample = c.encode(N(f["q_loc"], f["q_scale"]), N(f["p_loc"], f["p_scale"]), seed=int(f["seed"]))
    K = REF["tensor_K"]
    assert [len(i) for i in idx] == K.tolist()
    for r, ix in enumerate(idx):
        assert ix == REF["tensor_indices"][r, :K[r]].tolist(), r
    assert np.allclose(sample.cpu().numpy(), REF["tensor_sample"], rtol=0, atol=1e-5)
    dec = c.decode(N(f["p_loc"], f["p_scale"]), idx, seed=int(f["seed"]))
    assert np.allclose(dec.cpu().numpy(), REF["tensor_decoded"], rtol=0, atol=1e-5)


RAND = np.load(os.path.join(GOLDEN_DIR, "refpy_random_blocks.npz"))


def _rand_case(k):
    mq, sq, mp, sp = RAND[f"b{k}_in"]
    omega, eps1, B, seed, S = RAND[f"b{k}_meta"]
    return mq, sq, mp, sp, float(omega), float(eps1), int(B), int(seed), int(S)


@pytest.mark.both_suites
@pytest.mark.usefixtures("suite")
def test_oracle_equals_reference_python_on_48_random_blocks(oracle):
    """tests/golden/make_golden_refpy_random.py: six coder settings (S from 4 to 403, B from 7 to 30, S < B included), three
    statistics regimes, K up to ~100, dims 1..1000 -- what the reference's encode_block returned for each."""
    n_ok = 0
    for k in range(int(RAND["n"])):
        mq, sq, mp, sp, omega, eps1, B, seed, S = _rand_case(k)
        want = RAND[f"b{k}_indices"].tolist()
        ci, cs = oracle.encode_block(mq, sq, mp, sp, seed, omega, S, B, mode=oracle.CANONICAL, max_K=4096)
        li, ls = oracle.encode_block(mq, sq, mp, sp, seed, omega, S, B, mode=oracle.LITERAL, max_K=4096)
        assert ci == want and li == want, k
        assert np.allclose(cs, RAND[f"b{k}_sample"], rtol=0, atol=1e-5 * max(1.0, float(np.abs(cs).max()))), k
        assert np.allclose(RAND[f"b{k}_decoded"], RAND[f"b{k}_sample"], rtol=0, atol=1e-5 * max(1.0, float(np.abs(cs).max()))), k
        n_ok += 1
    assert n_ok == 48


@pytest.mark.gpu
def test_hip_equals_reference_python_on_48_random_blocks(engine):
    import irec
    import torch
    for k in range(int(RAND["n"])):
        mq, sq, mp, sp, omega, eps1, B, seed, S = _rand_case(k)
        c = irec.BeamSearchCoder(kl_per_partition=omega, n_beams=B, extra_samples=eps1)
        assert c.n_samples == S
        N = lambda a, b: torch.distributions.Normal(torch.as_tensor(a[None]).cuda(), torch.as_tensor(b[None]).cuda(), validate_args=False)  # noqa: E731
        idx, sample = c.encode(N(mq, sq), N(mp, sp), seed=seed)
        assert [int(i) for i in idx] == RAND[f"b{k}_indices"].tolist(), k
        got = sample.cpu().numpy().reshape(-1)
        assert np.allclose(got, RAND[f"b{k}_sample"], rtol=0, atol=1e-5 * max(1.0, float(np.abs(got).max()))), k


@pytest.mark.skipif(not os.path.isdir("/root/reference/rec/coding"), reason="live run needs the reference checkout (build container only)")
def test_reference_python_live_on_fresh_blocks(oracle):
    """The reference's encode_block / decode_block executed NOW (numpy TF stub) on random blocks of the four settings."""
    saved_path, saved_mods = list(sys.path), {k: sys.modules.get(k) for k in ("tensorflow", "tensorflow_probability")}
    sys.path[:0] = [os.path.join(ROOT, "oracle", "tfshim"), "/root/reference"]
    try:
        import tensorflow as tf
        import tensorflow_probability as tfp
        from rec.coding import BeamSearchCoder
        assert "/root/reference" in sys.modules[BeamSearchCoder.__module__].__file__
        rng = np.random.default_rng(2026)
        agree = 0
        for k, (omega, eps1, B) in enumerate([(3.0, 1.2, 20), (3.0, 1.0, 10), (5.0, 1.0, 30), (6.0, 1.0, 10), (3.0, 1.2, 20), (2.0, 1.0, 7)]):
            D = int(rng.choice([1000, 192, 57]))
            mq, sq, mp, sp = oracle.synthetic_latent(4000 + k, D)
            if k == 4:
                sq = (sq * 0.5).astype(np.float32)            # more partitions
            seed = int(rng.integers(0, 2 ** 31 - 200))
            coder = BeamSearchCoder(kl_per_partition=omega, n_beams=B, extra_samples=eps1)
            q = tfp.distributions.Normal(tf.constant(mq[None]), tf.constant(sq[None]))
            p = tfp.distributions.Normal(tf.constant(mp[None]), tf.constant(sp[None]))
            with contextlib.redirect_stdout(io.StringIO()):
                indices, sample = coder.encode_block(q, p, seed=seed)
                decoded = coder.decode_block(p, [int(i) for i in indices], seed=seed)
            oidx, osample = oracle.encode_block(mq, sq, mp, sp, seed, omega, coder.n_samples, B, mode=oracle.LITERAL)
            assert [int(i) for i in indices] == oidx, (k, omega, B, D)
            assert np.allclose(sample.numpy().reshape(-1), osample, rtol=0, atol=1e-5)
            assert np.allclose(decoded.numpy(), sample.numpy(), rtol=0, atol=1e-5)
            agree += 1
        assert agree == 6
    finally:
        sys.path[:] = saved_path
        for k in [m for m in sys.modules if m == "rec" or m.startswith("rec.") or m in ("tensorflow", "tensorflow_probability")]:
            del sys.modules[k]
        for k, v in saved_mods.items():
            if v is not None:
                sys.modules[k] = v
