"""rec/coding/utils.py: CodingError (:6) and stateless_gumbel_sample (:9-12)."""
import ctypes

import numpy as np

from ..errors import CodingError  # noqa: F401


def stateless_gumbel_sample(shape, seed):
    """-log(-log(tf.random.stateless_normal(shape, [seed, seed + 1]))) -- rec/coding/utils.py:9-12, as written (a NORMAL
    draw inside the double log: NaN wherever it falls outside (0, 1]).  Host numpy float32 of the given shape; the
    importance sampler's Gumbel-max branch (irec_importance_encode, alpha < inf) evaluates the same stream in C++."""
    from .. import _lib
    n = int(np.prod(shape))
    z = np.empty(n, dtype=np.float32)
    _lib.check(_lib.load().irec_tf_stateless_normal(int(seed), int(seed) + 1, n, z.ctypes.data_as(ctypes.c_void_p)),
               "irec_tf_stateless_normal")
    with np.errstate(invalid="ignore", divide="ignore"):
        return (-np.log(-np.log(z))).astype(np.float32).reshape(shape)
