"""Importance-sampler plumbing (BASELINE.json configs[0]; rec/coding/samplers.py:61-101, importance_sampling.py).

The reference runs this on the CPU, and its own test file for it is empty (rec/coding/tests/test_importance_sampling.py),
so nothing pins absolute indices: the host entry points of libirec_hip.so are checked against the oracle's restatement
(same libm), against float64 recomputation, and through the codec's own invariants.  No GPU involved.
"""
import math

import numpy as np
import pytest
import torch

import irec
from irec.coding import CodingError, ImportanceSampler

pytestmark = [pytest.mark.both_suites, pytest.mark.usefixtures("suite")]   # also run (as gpu-marked items) by the driver on the GPU box



def _normal(loc, scale):
    return torch.distributions.Normal(torch.as_tensor(loc), torch.as_tensor(scale), validate_args=False)


def test_tf_normal_stream_matches_oracle_and_is_standard_normal(oracle):
    import ctypes
    lib = irec._lib.load()
    for seed, n in [(42, 1001), (0, 7), (2 ** 31 - 1, 64), (123456789, 4096)]:
        got = np.zeros(n, dtype=np.float32)
        assert lib.irec_tf_random_normal(seed, n, got.ctypes.data_as(ctypes.c_void_p)) == 0
        assert np.array_equal(got, oracle.tf_random_normal(seed, n)), seed
    x = oracle.tf_random_normal(7, 200000).astype(np.float64)
    assert abs(x.mean()) < 0.01 and abs(x.var() - 1.0) < 0.02 and abs((x ** 3).mean()) < 0.03
    # prefix property the decoder relies on: sample(index + 1) is a prefix of sample(n_samples)
    assert np.array_equal(oracle.tf_random_normal(7, 13), x[:13].astype(np.float32))


def test_box_muller_against_float64(oracle):
    """Philox block -> (x0,x1),(x2,x3) -> Box-Muller with 23-bit uniforms."""
    seed = 99
    s2 = oracle.py_first_randint31(seed)
    got = oracle.tf_random_normal(seed, 8)
    for g in range(2):
        x = oracle.philox4x32([seed, 0], [g, 0, s2, 0]).astype(np.uint64)
        u = ((x & 0x7fffff).astype(np.float64)) / 2.0 ** 23
        for h in range(2):
            u1 = max(u[2 * h], 1e-7)
            r = math.sqrt(-2.0 * math.log(u1))
            v = 2.0 * math.pi * u[2 * h + 1]
            assert abs(got[4 * g + 2 * h] - math.sin(v) * r) < 2e-6
            assert abs(got[4 * g + 2 * h + 1] - math.cos(v) * r) < 2e-6


@pytest.mark.parametrize("n,bits,seed", [(1, 4.0, 1), (2, 7.5, 42), (7, 10.0, 3), (64, 8.0, 5), (300, 6.0, 11)])
def test_importance_sampler_matches_oracle_and_round_trips(oracle, n, bits, seed):
    rng = np.random.default_rng(n + seed)
    p_loc = rng.standard_normal(n).astype(np.float32)
    p_scale = np.exp(rng.normal(0, 0.25, n)).astype(np.float32)
    t_loc = (p_loc + p_scale * rng.normal(0, 0.5, n)).astype(np.float32)
    t_scale = (p_scale * np.exp(-np.abs(rng.normal(0, 0.3, n)))).astype(np.float32)
    sampler = ImportanceSampler(coding_bits=bits)
    S = sampler.n_samples()
    assert S == oracle.importance_n_samples(bits) and abs(S - 2.0 ** bits) <= 1.0 + 2.0 ** bits * 1e-6
    idx, sample = sampler.coded_sample(_normal(t_loc, t_scale), _normal(p_loc, p_scale), seed)
    ridx, rsample = oracle.importance_encode(t_loc, t_scale, p_loc, p_scale, bits, seed)
    assert idx == ridx and 0 <= idx < S
    assert np.array_equal(sample.numpy(), rsample)
    # decode(encode) is exact, and equals the oracle's decoder
    dec = sampler.decode_sample(_normal(p_loc, p_scale), idx, seed)
    assert torch.equal(dec, sample)
    assert np.array_equal(dec.numpy(), oracle.importance_decode(p_loc, p_scale, idx, seed))
    # the chosen proposal carries the largest importance weight (float64 recomputation; ties aside)
    x = oracle.tf_random_normal(seed, S * n).astype(np.float64).reshape(S, n)
    tl = (t_loc.astype(np.float64) - p_loc) / p_scale
    ts = t_scale.astype(np.float64) / p_scale
    w = (-0.5 * ((x - tl) / ts) ** 2 - np.log(ts) + 0.5 * x ** 2).sum(axis=1)
    assert w[idx] >= w.max() - 1e-3 * max(1.0, abs(w.max()))
    assert np.allclose(sample.numpy(), p_scale * x[idx] + p_loc, atol=1e-5)
    assert sampler.get_codelength(idx) == pytest.approx(bits * math.log(2.0), rel=1e-6)


def test_importance_sampler_shapes_and_errors():
    p = _normal(np.zeros((1, 2, 3), np.float32), np.ones((1, 2, 3), np.float32))
    t = _normal(np.full((1, 2, 3), 0.3, np.float32), np.full((1, 2, 3), 0.5, np.float32))
    s = ImportanceSampler(coding_bits=5)
    idx, x = s.coded_sample(t, p, seed=9)
    assert x.shape == (1, 2, 3) and x.dtype == torch.float32
    assert torch.equal(s.decode_sample(p, idx, seed=9), x)
    with pytest.raises(CodingError):
        ImportanceSampler(coding_bits=5, alpha=0.5).coded_sample(t, p, seed=9)      # importance_sampling.py:33-34
    with pytest.raises(CodingError):
        ImportanceSampler(coding_bits=5, alpha=float("nan")).coded_sample(t, p, seed=9)
    with pytest.raises(irec._lib.IrecLibraryError):
        ImportanceSampler(coding_bits=40).coded_sample(t, p, seed=9)                # 2^40 proposals: refused


# ---- Gumbel-max branch (importance_sampling.py:67-71, rec/coding/utils.py:9-12) ---------------------------------------------
def test_stateless_normal_stream_key_scramble(oracle):
    """tf.random.stateless_normal(seed=[s0, s1]): one Philox block under the fixed key over the seed pair gives the stream's
    key and upper counter half (stateless_random_ops.cc GenerateKey); product == oracle, and the layout is re-derived here."""
    import ctypes
    for s0, s1 in [(1, 2), (43, 44), (2 ** 31, 2 ** 31 + 1), (123456789012, 5)]:
        got = np.empty(11, np.float32)
        irec._lib.check(irec._lib.load().irec_tf_stateless_normal(s0, s1, 11, got.ctypes.data_as(ctypes.c_void_p)), "stateless")
        assert np.array_equal(got, oracle.tf_stateless_normal(s0, s1, 11))
        mix = oracle.philox4x32([0x3ec8f720, 0x02461e29], [s0 & 0xFFFFFFFF, s0 >> 32, s1 & 0xFFFFFFFF, s1 >> 32])
        x = oracle.philox4x32([mix[0], mix[1]], [1, 0, mix[2], mix[3]]).astype(np.uint64)      # block 1 -> elements 4..7
        u = (x & 0x7fffff).astype(np.float64) / 2.0 ** 23
        r = math.sqrt(-2.0 * math.log(max(u[0], 1e-7)))
        assert abs(got[4] - math.sin(2 * math.pi * u[1]) * r) < 2e-6 and abs(got[5] - math.cos(2 * math.pi * u[1]) * r) < 2e-6
    z = oracle.tf_stateless_normal(7, 8, 40000).astype(np.float64)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02


def test_stateless_gumbel_sample_as_written():
    from irec.coding.utils import stateless_gumbel_sample
    g = stateless_gumbel_sample((4, 5), 10)
    assert g.shape == (4, 5) and g.dtype == np.float32
    # a normal draw lies in (0, 1] about a third of the time; everywhere else the double log is NaN (reference quirk kept)
    g = stateless_gumbel_sample((4000,), 3)
    assert 0.25 < np.isfinite(g).mean() < 0.45


@pytest.mark.parametrize("n,bits,seed,alpha", [(1, 6.0, 1, 1.0), (5, 8.0, 42, 1.0), (16, 9.0, 7, 2.5), (64, 7.0, 5, 1.0),
                                               (3, 10.0, 11, 30.0)])
def test_gumbel_max_branch_matches_oracle(oracle, n, bits, seed, alpha):
    rng = np.random.default_rng(1000 * n + seed)
    p_loc = rng.standard_normal(n).astype(np.float32)
    p_scale = np.exp(rng.normal(0, 0.25, n)).astype(np.float32)
    t_loc = (p_loc + p_scale * rng.normal(0, 0.5, n)).astype(np.float32)
    t_scale = (p_scale * np.exp(-np.abs(rng.normal(0, 0.3, n)))).astype(np.float32)
    sampler = ImportanceSampler(coding_bits=bits, alpha=alpha)
    S = sampler.n_samples()
    idx, sample = sampler.coded_sample(_normal(t_loc, t_scale), _normal(p_loc, p_scale), seed)
    ridx, rsample = oracle.importance_encode(t_loc, t_scale, p_loc, p_scale, bits, seed, alpha=alpha)
    assert idx == ridx and 0 <= idx < S and np.array_equal(sample.numpy(), rsample)
    assert torch.equal(sampler.decode_sample(_normal(p_loc, p_scale), idx, seed), sample)     # the decoder ignores alpha
    # float64 recomputation: the winner maximises alpha * w + g among the proposals whose perturbation is finite
    x = oracle.tf_random_normal(seed, S * n).astype(np.float64).reshape(S, n)
    tl = (t_loc.astype(np.float64) - p_loc) / p_scale
    ts = t_scale.astype(np.float64) / p_scale
    w = (-0.5 * ((x - tl) / ts) ** 2 - np.log(ts) + 0.5 * x ** 2).sum(axis=1)
    z = oracle.tf_stateless_normal(seed + 1, seed + 2, S).astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        pert = alpha * w - np.log(-np.log(z))
    finite = np.isfinite(pert) | (pert == np.inf)
    assert finite[idx] and pert[idx] >= np.nanmax(np.where(finite, pert, -np.inf)) - 1e-3 * max(1.0, abs(pert[idx]))
    # with a large alpha the perturbation cannot overturn a clear winner among the finite ones; alpha = inf ignores it
    i_inf, _ = ImportanceSampler(coding_bits=bits).coded_sample(_normal(t_loc, t_scale), _normal(p_loc, p_scale), seed)
    assert i_inf == int(np.argmax(w)) or abs(w[i_inf] - w.max()) < 1e-3
