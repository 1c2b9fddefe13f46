"""Stub `tensorflow_probability`: tfd.Normal (quantile / log_prob / sample) and kl_divergence with TFP 0.9's float32 formulas
(SURVEY.md A4).  Test infrastructure -- see ../README.md."""
import types

import numpy as np
import tensorflow as tf

from oracle import oracle as _O

_LUT = None


def _ndtri_f32(p):
    """TFP special_math.ndtri in float32.  The reference only evaluates it at p = float32(k) / 10007 (beam_search_coder.py:45-49):
    those come from the oracle's table; anything else goes through the oracle's scalar restatement."""
    global _LUT
    if _LUT is None:
        _LUT = _O.build_lut()
    p = np.asarray(p, dtype=np.float32)
    k = np.rint(p.astype(np.float64) * 10007.0).astype(np.int64)
    ok = (k >= 1) & (k <= 10006)
    kk = np.where(ok, k, 1)
    on_grid = ok & ((kk.astype(np.float32) / np.float32(10007)) == p)
    out = _LUT[kk].astype(np.float32)
    if not on_grid.all():
        f = _O.lib().irec_oracle_ndtri_f32
        flat, og = out.reshape(-1), on_grid.reshape(-1)
        for i in np.nonzero(~og)[0]:
            flat[i] = f(float(p.reshape(-1)[i]))
        out = flat.reshape(p.shape)
    return out


class Normal:
    def __init__(self, loc, scale, **kwargs):
        self.loc = loc if isinstance(loc, tf.Tensor) else tf.constant(loc)
        self.scale = scale if isinstance(scale, tf.Tensor) else tf.constant(scale)

    def quantile(self, p):
        return tf.Tensor(_ndtri_f32(tf._np(p))) * self.scale + self.loc          # _inv_z(ndtri(p))

    def log_prob(self, x):
        x = x if isinstance(x, tf.Tensor) else tf.constant(x)
        log_unnormalized = -0.5 * tf.math.squared_difference(x / self.scale, self.loc / self.scale)
        log_normalization = tf.constant(np.float32(0.5 * np.log(2. * np.pi))) + tf.math.log(self.scale)
        return log_unnormalized - log_normalization

    def sample(self, n=None, seed=None):
        shp = ([] if n is None else [int(n)]) + list(np.broadcast(tf._np(self.loc), tf._np(self.scale)).shape)
        return tf.random.normal(shp) * self.scale + self.loc


def kl_divergence(a, b):
    diff_log_scale = tf.math.log(a.scale) - tf.math.log(b.scale)
    return (0.5 * tf.math.squared_difference(a.loc / b.scale, b.loc / b.scale) +
            0.5 * tf.math.expm1(2. * diff_log_scale) - diff_log_scale)


distributions = types.SimpleNamespace(Normal=Normal, kl_divergence=kl_divergence, Distribution=object)
