"""48 random blocks through the REFERENCE'S OWN PYTHON (see make_golden_refpy.py for how it runs here): inputs and what
rec.coding.BeamSearchCoder.encode_block / decode_block returned -> tests/golden/refpy_random_blocks.npz.
Build container only.  Run:  python tests/golden/make_golden_refpy_random.py"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle", "tfshim"), ROOT, os.path.join(ROOT, "scripts"), "/root/reference"]

import tensorflow as tf                      # noqa: E402  (the stub)
import tensorflow_probability as tfp         # noqa: E402  (the stub)
from rec.coding import BeamSearchCoder       # noqa: E402  (the REAL reference class)
from margins import random_block             # noqa: E402
from oracle import oracle as O               # noqa: E402

SETTINGS = [(3.0, 1.2, 20), (3.0, 1.0, 10), (5.0, 1.0, 30), (6.0, 1.0, 10), (2.0, 1.5, 7), (1.5, 1.0, 20)]


def main():
    rng = np.random.default_rng(424242)
    out = {"kind": "refpy_random", "n": 48}
    agree = 0
    for k in range(48):
        omega, eps1, B = SETTINGS[k % len(SETTINGS)]
        D = int(rng.choice([1000, 192, 64, int(rng.integers(1, 513))])) if k % 8 else 1000
        mq, sq, mp, sp = random_block(rng, D, k % 3)
        seed = int(rng.integers(0, 2 ** 31 - 1000))
        coder = BeamSearchCoder(kl_per_partition=omega, n_beams=B, extra_samples=eps1)
        q = tfp.distributions.Normal(tf.constant(mq[None]), tf.constant(sq[None]))
        p = tfp.distributions.Normal(tf.constant(mp[None]), tf.constant(sp[None]))
        with contextlib.redirect_stdout(io.StringIO()):
            indices, sample = coder.encode_block(q, p, seed=seed)
            indices = [int(i) for i in indices]
            decoded = coder.decode_block(p, list(indices), seed=seed)
        oidx, _ = O.encode_block(mq, sq, mp, sp, seed, omega, coder.n_samples, B, mode=O.CANONICAL)
        agree += int(oidx == indices)
        out[f"b{k}_in"] = np.stack([mq, sq, mp, sp])
        out[f"b{k}_meta"] = np.array([omega, eps1, B, seed, coder.n_samples], np.float64)
        out[f"b{k}_indices"] = np.array(indices, np.int32)
        out[f"b{k}_sample"] = sample.numpy().reshape(-1).astype(np.float32)
        out[f"b{k}_decoded"] = decoded.numpy().reshape(-1).astype(np.float32)
        print(f"block {k}: D={D} S={coder.n_samples} B={B} K={len(indices)} oracle==reference: {oidx == indices}", flush=True)
    out["oracle_canonical_agreed_at_generation"] = agree
    np.savez_compressed(os.path.join(HERE, "refpy_random_blocks.npz"), **out)
    print(f"wrote refpy_random_blocks.npz; oracle (canonical) == reference python on {agree} / 48")


if __name__ == "__main__":
    main()
