"""Runs the REFERENCE'S OWN PYTHON (rec.coding from /root/reference, imported unmodified) on the committed fixtures, with the
TensorFlow / TFP calls it makes served by the numpy stubs of oracle/tfshim (primitives from the C oracle; see its README), and
commits what it returns as tests/golden/refpy_*.npz.

Build container only (/root/reference does not exist on the GPU box; the vectors travel, the reference does not).
Run:  python tests/golden/make_golden_refpy.py
"""
import contextlib
import glob
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle", "tfshim"), ROOT, "/root/reference"]

import tensorflow as tf                      # noqa: E402  (the stub)
import tensorflow_probability as tfp         # noqa: E402  (the stub)
from rec.coding import BeamSearchCoder       # noqa: E402  (the REAL reference class)
from rec.coding.importance_sampling import (decode_gaussian_importance_sample,   # noqa: E402
                                            encode_gaussian_importance_sample)

tfd = tfp.distributions
assert BeamSearchCoder.__module__ == "rec.coding.beam_search_coder" and "/root/reference" in sys.modules[BeamSearchCoder.__module__].__file__


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):      # the reference prints the KL of every block
        return fn(*a, **k)


def main():
    out = {"kind": "refpy", "note": "outputs of /root/reference/rec/coding run with oracle/tfshim (numpy TF stub); see tests/golden/make_golden_refpy.py"}
    names = []
    for path in sorted(glob.glob(os.path.join(HERE, "block_*.npz")) + [os.path.join(HERE, "ref_test_beam_search.npz")]):
        g = np.load(path)
        name = os.path.basename(path)[:-4]
        coder = BeamSearchCoder(kl_per_partition=float(g["kl_per_partition"]), n_beams=int(g["n_beams"]),
                                extra_samples=float(g["extra_samples"]))
        assert coder.n_samples == int(g["n_samples"])
        q = tfd.Normal(loc=tf.constant(g["q_loc"][None]), scale=tf.constant(g["q_scale"][None]))
        p = tfd.Normal(loc=tf.constant(g["p_loc"][None]), scale=tf.constant(g["p_scale"][None]))
        indices, sample = quiet(coder.encode_block, q, p, seed=int(g["seed"]))
        indices = [int(i) for i in indices]
        decoded = quiet(coder.decode_block, p, list(indices), seed=int(g["seed"]))
        out[f"{name}_indices"] = np.array(indices, np.int32)
        out[f"{name}_sample"] = sample.numpy().reshape(-1).astype(np.float32)
        out[f"{name}_decoded"] = decoded.numpy().reshape(-1).astype(np.float32)
        out[f"{name}_codelength"] = np.float64(coder.get_codelength(indices))
        names.append(name)
        print(f"{name}: K = {len(indices)}  same as the oracle fixture: {indices == g['indices'].tolist()}", flush=True)
    out["block_names"] = np.array(names)

    # GaussianCoder.encode / decode with block_size: the reference's split -> per-block loop -> merge (coder.py:412-491)
    g = np.load(os.path.join(HERE, "tensor_rvae_cfg2.npz"))
    coder = BeamSearchCoder(kl_per_partition=float(g["kl_per_partition"]), n_beams=int(g["n_beams"]),
                            extra_samples=float(g["extra_samples"]), block_size=int(g["block_size"]))
    q = tfd.Normal(loc=tf.constant(g["q_loc"]), scale=tf.constant(g["q_scale"]))
    p = tfd.Normal(loc=tf.constant(g["p_loc"]), scale=tf.constant(g["p_scale"]))
    indices, sample = quiet(coder.encode, q, p, seed=int(g["seed"]))
    decoded = quiet(coder.decode, p, [list(ix) for ix in indices], seed=int(g["seed"]))
    K = np.array([len(ix) for ix in indices], np.int32)
    flat = np.full((len(indices), K.max()), -1, np.int32)
    for r, ix in enumerate(indices):
        flat[r, :len(ix)] = [int(v) for v in ix]
    out["tensor_K"], out["tensor_indices"] = K, flat
    out["tensor_sample"] = sample.numpy().astype(np.float32)
    out["tensor_decoded"] = decoded.numpy().astype(np.float32)
    print("tensor_rvae_cfg2: K per block", K.tolist(), " same as the oracle fixture:", np.array_equal(flat, g["indices"]))

    # importance sampler (config 1 plumbing): both branches of encode_gaussian_importance_sample (importance_sampling.py:9-103)
    rng = np.random.default_rng(11)
    cases = []
    for n, bits, seed, alpha in [(1, 4.0, 1, float("inf")), (7, 8.0, 3, float("inf")), (64, 6.0, 5, float("inf")),
                                 (5, 8.0, 42, 1.0), (16, 7.0, 7, 2.5)]:
        p_loc = rng.standard_normal(n).astype(np.float32)
        p_scale = np.exp(rng.normal(0, 0.25, n)).astype(np.float32)
        t_loc = (p_loc + p_scale * rng.normal(0, 0.5, n)).astype(np.float32)
        t_scale = (p_scale * np.exp(-np.abs(rng.normal(0, 0.3, n)))).astype(np.float32)
        idx, samp = encode_gaussian_importance_sample(tf.constant(t_loc), tf.constant(t_scale), tf.constant(p_loc),
                                                      tf.constant(p_scale), coding_bits=bits, seed=seed, alpha=alpha)
        dec = decode_gaussian_importance_sample(tf.constant(p_loc), tf.constant(p_scale), idx, seed)
        c = len(cases)
        out[f"is{c}_in"] = np.stack([t_loc, t_scale, p_loc, p_scale])
        out[f"is{c}_meta"] = np.array([bits, seed, alpha], np.float64)
        out[f"is{c}_index"] = np.int64(int(idx))
        out[f"is{c}_sample"] = samp.numpy().astype(np.float32)
        out[f"is{c}_decoded"] = dec.numpy().astype(np.float32)
        cases.append(c)
        print(f"importance case {c}: n={n} bits={bits} alpha={alpha} -> index {int(idx)}")
    out["n_importance_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, "refpy_reference_outputs.npz"), **out)
    print("wrote refpy_reference_outputs.npz")


if __name__ == "__main__":
    main()
