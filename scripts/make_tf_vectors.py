#!/usr/bin/env python3
"""Harvest golden vectors from the REAL reference stack (TensorFlow 2.1.0 + TensorFlow-Probability 0.9.0 + the
reference's own rec.coding) -- the only thing that can move this repo's oracle from "parity unpinned" to pinned.

Runs ONLY where those packages exist (python 3.6/3.7, `pip install tensorflow==2.1.0 tensorflow-probability==0.9.0`,
a checkout of gergely-flamich/relative-entropy-coding); it cannot run in the build image (no TF wheel for cp310, no
network) and it never runs on the GPU box.  It imports the reference, it does not contain any of its code.

    python scripts/make_tf_vectors.py --reference /path/to/relative-entropy-coding [--out tests/golden]

writes tests/golden/tf_primitives.npz and tests/golden/tf_encode_blocks.npz; commit them.  tests/test_tf_vectors.py
consumes them when present (and says so loudly when they are not).  Every array is DATA (inputs and what TF returned).

What is dumped, and which assumption of SURVEY.md Appendix A it pins:
  A1/A2  uniform_<seed>_<S>x<D>   tf.random.set_seed(seed); tf.random.uniform([S,1,D], 1, 10007, seed=seed, dtype=int32)
                                  (beam_search_coder.py:38-43): Philox key/counter layout, seed pair, `1 + u32 % 10006`
  A1/A5  shuffle_<seed>_<n>       tf.random.set_seed(seed); tf.random.shuffle(tf.range(n))   (coder.py:62-64)
  A4     quantile                 tfd.Normal(0,1).quantile(float32(k)/float32(10007)), k = 1..10006 (float32)
  A4     log_prob_*, kl_*         tfd.Normal.log_prob / tfd.kl_divergence on a fixed grid
  A3     argsort_ties             tf.argsort(v, direction='DESCENDING') on a vector with ties, NaN-free
  A6     normal_<seed>_<n>        tf.random.set_seed(seed); tfd.Normal(0,1).sample(n)   (importance_sampling.py:37,50-53)
  A7     reduce_sum_*             tf.reduce_sum over D of a fixed float32 vector (Eigen's order: documents the noise)
  a7     encode_<fixture>         BeamSearchCoder.encode_block + decode_block on every tests/golden/block_*.npz input
                                  (indices, sample, K): the end-to-end pin of the hot path
"""
import argparse
import glob
import os
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of gergely-flamich/relative-entropy-coding")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    args = ap.parse_args()
    os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")          # the reference codes on the CPU (compression_performance.py:16)
    import tensorflow as tf
    import tensorflow_probability as tfp
    tfd = tfp.distributions
    if not tf.__version__.startswith("2.1"):
        print(f"WARNING: TensorFlow {tf.__version__}, the reference pins 2.1.0 (requirements.txt:9)", file=sys.stderr)
    sys.path.insert(0, args.reference)
    from rec.coding import BeamSearchCoder               # the reference's own class

    seeds = [0, 42, 69420, 2 ** 31 - 1]
    prim = {"tf_version": tf.__version__, "tfp_version": tfp.__version__, "seeds": np.array(seeds, np.int64)}
    for seed in seeds:
        for S, D in ((36, 1000), (36, 192), (20, 7), (5, 1), (148, 3)):
            tf.random.set_seed(seed)
            prim[f"uniform_{seed}_{S}x{D}"] = tf.random.uniform([S, 1, D], 1, 10007, seed=seed, dtype=tf.int32).numpy()
        # the encoder seeds step t with seed + t: a few consecutive seeds pin the "+ t" plumbing
        for t in (1, 2, 7):
            tf.random.set_seed(seed + t)
            prim[f"uniform_{seed}+{t}_36x16"] = tf.random.uniform([36, 1, 16], 1, 10007, seed=seed + t, dtype=tf.int32).numpy()
        for n in (2, 10, 1000, 8192):
            tf.random.set_seed(seed)
            prim[f"shuffle_{seed}_{n}"] = tf.random.shuffle(tf.range(n)).numpy()
        for n in (1, 2, 7, 64, 1000):
            tf.random.set_seed(seed)
            prim[f"normal_{seed}_{n}"] = tfd.Normal(loc=tf.zeros([n]), scale=tf.ones([n])).sample(3).numpy()
            tf.random.set_seed(seed)
            prim[f"tfnormal_{seed}_{n}"] = tf.random.normal([3, n]).numpy()
    k = np.arange(1, 10007, dtype=np.int32)
    u = tf.cast(k, tf.float32) / 10007.
    prim["quantile_u"] = u.numpy()
    prim["quantile"] = tfd.Normal(loc=tf.zeros([]), scale=tf.ones([])).quantile(u).numpy()
    prim["quantile_scaled"] = tfd.Normal(loc=tf.zeros([]), scale=tf.constant(0.37)).quantile(u).numpy()
    rng = np.random.default_rng(7)
    x = rng.normal(0, 2, 4096).astype(np.float32)
    loc = rng.normal(0, 1, 4096).astype(np.float32)
    scale = np.exp(rng.normal(0, 0.5, 4096)).astype(np.float32)
    prim["grid_x"], prim["grid_loc"], prim["grid_scale"] = x, loc, scale
    prim["log_prob"] = tfd.Normal(loc, scale).log_prob(x).numpy()
    loc2 = rng.normal(0, 1, 4096).astype(np.float32)
    scale2 = np.exp(rng.normal(0, 0.5, 4096)).astype(np.float32)
    prim["grid_loc2"], prim["grid_scale2"] = loc2, scale2
    prim["kl_per_dim"] = tfd.kl_divergence(tfd.Normal(loc, scale), tfd.Normal(loc2, scale2)).numpy()
    prim["kl_sum"] = tf.reduce_sum(tfd.kl_divergence(tfd.Normal(loc, scale), tfd.Normal(loc2, scale2))).numpy()
    v = np.array([3., 1., 3., -0., 0., 7., 1., 3., -2., 7.], np.float32)
    prim["argsort_ties_in"] = v
    prim["argsort_ties"] = tf.argsort(v, direction='DESCENDING').numpy()
    w = rng.normal(0, 1, (64, 1000)).astype(np.float32)
    prim["reduce_sum_in"] = w
    prim["reduce_sum"] = tf.reduce_sum(w, axis=1).numpy()
    prim["floormod"] = tf.math.floormod(tf.constant([-7, -1, 0, 5, 10006, 2 ** 31 - 1], tf.int32), 10006).numpy()
    np.savez_compressed(os.path.join(args.out, "tf_primitives.npz"), **prim)
    print("wrote tf_primitives.npz:", len(prim), "arrays")

    enc = {"tf_version": tf.__version__, "tfp_version": tfp.__version__}
    names = []
    for path in sorted(glob.glob(os.path.join(args.out, "block_*.npz")) + glob.glob(os.path.join(args.out, "ref_test_beam_search.npz"))):
        g = np.load(path)
        name = os.path.basename(path)[:-4]
        coder = BeamSearchCoder(kl_per_partition=float(g["kl_per_partition"]), n_beams=int(g["n_beams"]),
                                extra_samples=float(g["extra_samples"]))
        q = tfd.Normal(loc=g["q_loc"][None], scale=g["q_scale"][None])
        p = tfd.Normal(loc=g["p_loc"][None], scale=g["p_scale"][None])
        try:
            indices, sample = coder.encode_block(q, p, seed=int(g["seed"]))
        except NameError:      # KL == 0: the reference leaves `beams` undefined (beam_search_coder.py:66,118)
            continue
        indices = [int(i) for i in indices]
        decoded = coder.decode_block(p, list(indices), seed=int(g["seed"]))
        enc[f"{name}_indices"] = np.array(indices, np.int32)
        enc[f"{name}_sample"] = np.asarray(sample, np.float32).reshape(-1)
        enc[f"{name}_decoded"] = np.asarray(decoded, np.float32).reshape(-1)
        enc[f"{name}_kl"] = np.float32(tf.reduce_sum(tfd.kl_divergence(q, p)).numpy())
        names.append(name)
        print(f"{name}: K = {len(indices)}")
    enc["names"] = np.array(names)
    np.savez_compressed(os.path.join(args.out, "tf_encode_blocks.npz"), **enc)
    print("wrote tf_encode_blocks.npz:", len(names), "fixtures")


if __name__ == "__main__":
    main()
