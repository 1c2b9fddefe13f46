"""CPU tests of the oracle (oracle/irec_oracle.c): known-answer vectors, golden fixtures, codec invariants."""
import math
import os
import random

import numpy as np
import pytest

from conftest import golden_files


def test_philox_random123_kats(oracle):
    # Random123 Philox4x32-10 known-answer vectors
    assert [hex(x) for x in oracle.philox4x32([0, 0], [0, 0, 0, 0])] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(x) for x in oracle.philox4x32([0xffffffff] * 2, [0xffffffff] * 4)] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(x) for x in oracle.philox4x32([0xa4093822, 0x299f31d0], [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344])] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_uniform_int_range_and_layout(oracle):
    r = oracle.uniform_int(42, 36 * 1000)
    assert r.min() >= 1 and r.max() <= 10006
    # element e uses Philox block e >> 2, lane e & 3 with key = seed, counter = (blk, 0, seed, 0)
    for e in (0, 1, 5, 1002, 35999):
        blk = oracle.philox4x32([42, 0], [e >> 2, 0, 42, 0])
        assert r[e] == 1 + int(blk[e & 3]) % 10006
    # a prefix of a longer draw is the shorter draw (stream is position-addressed)
    assert np.array_equal(oracle.uniform_int(42, 100), r[:100])
    # seeds are truncated mod 2^31 - 1; (0, 0) -> (0, 2^31 - 1)
    assert np.array_equal(oracle.uniform_int(42 + (2 ** 31 - 1), 64), r[:64])
    z = oracle.uniform_int(0, 8)
    assert z[0] == 1 + int(oracle.philox4x32([0, 0], [0, 0, 2 ** 31 - 1, 0])[0]) % 10006


def test_det_log_accuracy(oracle):
    rng = np.random.default_rng(0)
    for x in np.exp(rng.uniform(-700, 700, 5000)).tolist() + [1.0, 2.0, 0.5, 1.4142135623730951, 5e-324, 1e-310]:
        ref = math.log(x)
        assert abs(oracle.det_log(x) - ref) <= 4e-16 * max(abs(ref), 1.0), x
    assert oracle.det_log(1.0) == 0.0


def test_lut_against_scipy(oracle):
    from scipy.special import ndtri
    lut = oracle.build_lut()
    k = np.arange(1, 10007)
    p32 = k.astype(np.float32) / np.float32(10007)
    ref = ndtri(p32.astype(np.float64))
    assert np.abs(lut[1:] - ref).max() < 6e-7          # float32 Cephes evaluation error (SURVEY: 4.7e-7)
    assert np.all(np.diff(lut[1:]) > 0)                # strictly increasing quantiles
    assert lut[0] == 0.0
    assert abs(lut[1] + 3.7191932) < 1e-6 and abs(lut[10006] - 3.7191253) < 1e-6


def test_python_mt_seed_plumbing(oracle):
    for s in [0, 1, 42, 69420, 2 ** 31, 2 ** 40 + 5, -7, 2 ** 63 - 1]:
        assert oracle.py_first_randint31(s) == random.Random(s).randint(0, 2 ** 31 - 1)


def test_shuffle_is_a_permutation_and_fisher_yates(oracle):
    for seed, n in [(42, 8192), (0, 10), (7, 1), (5, 2), (69420, 12288)]:
        p = oracle.tf_shuffle_perm(seed, n)
        assert sorted(p.tolist()) == list(range(n))
    # replay Fisher-Yates by hand from the raw Philox stream
    seed, n = 42, 50
    op_seed = random.Random(seed).randint(0, 2 ** 31 - 1)
    a = list(range(n))
    for i in range(n - 1):
        blk = oracle.philox4x32([seed, 0], [i >> 2, 0, op_seed % (2 ** 31 - 1), 0])
        j = i + int(blk[i & 3]) % (n - i)
        a[i], a[j] = a[j], a[i]
    assert a == oracle.tf_shuffle_perm(seed, n).tolist()


def test_simple_hash(oracle):
    # floormod(sum idx[j] * (69 + j), 10006) + 1, empty path -> 1   (beam_search_coder.py:33-35)
    assert oracle.simple_hash([]) == 1
    assert oracle.simple_hash([5]) == (5 * 69) % 10006 + 1
    assert oracle.simple_hash([1, 2, 3]) == (69 + 140 + 213) % 10006 + 1
    big = [402] * 700
    s = sum(402 * (69 + j) for j in range(700))
    assert oracle.simple_hash(big) == ((s + 2 ** 31) % 2 ** 32 - 2 ** 31) % 10006 + 1


def test_aux_ratio(oracle):
    for i in range(20):
        assert oracle.aux_ratio(i) == np.float32(np.power(i + 1., -0.7864636765648174))


def test_reference_unit_test_round_trip(oracle):
    # rec/coding/tests/test_coder.py:12-21
    S = oracle.n_samples(6., 1.)
    assert S == 403
    idx, sample = oracle.encode_block([5.1], [0.001], [0.], [1.], 69420, 6., S, 10)
    assert len(idx) == 4 and all(0 <= i < S for i in idx)
    rec = oracle.decode_block([0.], [1.], idx, 69420, S)
    assert np.array_equal(rec, sample)
    assert abs(float(sample[0]) - 5.1) < 0.01
    idx_l, sample_l = oracle.encode_block([5.1], [0.001], [0.], [1.], 69420, 6., S, 10, mode=oracle.LITERAL)
    assert idx_l == idx and np.array_equal(sample_l, sample)


@pytest.mark.parametrize("path", golden_files("block"), ids=os.path.basename)
def test_golden_blocks(oracle, path):
    g = np.load(path)
    S, B = int(g["n_samples"]), int(g["n_beams"])
    assert S == oracle.n_samples(float(g["kl_per_partition"]), float(g["extra_samples"]))
    idx, sample, tr = oracle.encode_block(g["q_loc"], g["q_scale"], g["p_loc"], g["p_scale"], int(g["seed"]),
                                          float(g["kl_per_partition"]), S, B, trace=True)
    assert idx == g["indices"].tolist()
    assert np.array_equal(sample, g["sample"])
    assert np.array_equal(tr["sel"], g["sel"])
    assert np.float32(oracle.block_kl(g["q_loc"], g["q_scale"], g["p_loc"], g["p_scale"])) == g["kl"]
    assert len(idx) == int(g["K"]) and all(0 <= i < S for i in idx)
    dec = oracle.decode_block(g["p_loc"], g["p_scale"], idx, int(g["seed"]), S)
    assert np.array_equal(dec, sample)                      # decode(encode) bit exact
    assert abs(oracle.codelength(idx, S) - float(g["codelength"])) < 1e-9
    # literal (TF op order) mode picks the same indices on every fixture
    idx_l, _ = oracle.encode_block(g["q_loc"], g["q_scale"], g["p_loc"], g["p_scale"], int(g["seed"]),
                                   float(g["kl_per_partition"]), S, B, mode=oracle.LITERAL)
    assert idx_l == g["indices_literal"].tolist() == idx


def test_golden_tensor(oracle):
    g = np.load(golden_files("tensor")[0])
    S, B, bs = int(g["n_samples"]), int(g["n_beams"]), int(g["block_size"])
    idx, sample = oracle.encode_tensor(g["q_loc"], g["q_scale"], g["p_loc"], g["p_scale"], int(g["seed"]),
                                       float(g["kl_per_partition"]), S, B, block_size=bs)
    assert [len(i) for i in idx] == g["K"].tolist()
    for r, i in enumerate(idx):
        assert i == g["indices"][r, :len(i)].tolist()
    assert np.array_equal(sample, g["sample"])
    dec = oracle.decode_tensor(g["p_loc"], g["p_scale"], idx, int(g["seed"]), S, block_size=bs)
    assert np.array_equal(dec, sample)
    assert np.array_equal(oracle.tf_shuffle_perm(int(g["seed"]), 8192)[:64], g["perm_head"])


def test_importance_weight_tracks_kl(oracle):
    # codec invariant: log q(z)/p(z) at the coded sample is of the order of the KL
    mq, sq, mp, sp = oracle.synthetic_latent(3, 1000)
    idx, z = oracle.encode_block(mq, sq, mp, sp, 42, 3., 36, 20)
    lw = np.sum(-0.5 * ((z - mq) / sq) ** 2 - np.log(sq) + 0.5 * ((z - mp) / sp) ** 2 + np.log(sp))
    kl = oracle.block_kl(mq, sq, mp, sp)
    assert 0.5 * kl < lw < 1.5 * kl


def test_k_zero_block(oracle):
    # q == p: KL = 0 -> no partitions.  (Reference: NameError at beam_search_coder.py:118; here: sample = p.loc.)
    mp = np.float32([0.3, -1.0]); sp = np.float32([1.0, 2.0])
    idx, sample = oracle.encode_block(mp, sp, mp, sp, 1, 3., 20, 10)
    assert idx == [] and np.array_equal(sample, mp)
    assert np.array_equal(oracle.decode_block(mp, sp, [], 1, 20), mp)
