"""Host mirror of rec/coding/samplers.py for the importance sampler (BASELINE.json configs[0] plumbing).

`ImportanceSampler` keeps the reference's constructor and method surface (samplers.py:61-101).  The reference runs it
on the CPU -- e^KL standard-normal proposals from TensorFlow's global Philox stream, an argmax over their importance
weights (importance_sampling.py:9-103) -- and so does this mirror: it calls the host-side entry points
`irec_importance_encode / _decode` of libirec_hip.so (include/irec.h), which need no GPU.  It is not on the
`sampler='beam_search'` hot path and no device kernel is involved.  Both branches are built: `alpha = inf` (the
reference's default, the setting its models use: resnet_vae.py:126-131) takes the argmax of the importance weights,
`1 <= alpha < inf` the Gumbel-max of importance_sampling.py:67-71 over `stateless_gumbel_sample`
(rec/coding/utils.py:9-12).  The rejection sampler is out of scope.
"""
import abc
import ctypes
import math  # noqa: F401

import numpy as np
import torch

from .. import _lib
from .utils import CodingError


class Sampler(abc.ABC):
    """samplers.py:15-58."""

    def __init__(self, name="sampler", **kwargs):
        self.name = name

    @abc.abstractmethod
    def coded_sample(self, target, coder, seed):
        """-> (sample index, sample)"""

    @abc.abstractmethod
    def decode_sample(self, coder, sample_index, seed):
        """-> the sample with the given index"""

    @abc.abstractmethod
    def get_codelength(self, index):
        pass

    @abc.abstractmethod
    def update(self, target, coder):
        pass


def _host_f32(t):
    return np.ascontiguousarray(torch.as_tensor(t).detach().to("cpu", torch.float32).numpy()).reshape(-1)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class ImportanceSampler(Sampler):
    """samplers.py:61-101."""

    def __init__(self, coding_bits, alpha=np.inf, name="importance_sampler", **kwargs):
        super().__init__(name=name, **kwargs)
        self.alpha = alpha
        self.coding_bits = coding_bits

    def _check_alpha(self):
        if not self.alpha >= 1.:   # importance_sampling.py:33-34
            raise CodingError(f"Alpha must be in the range [1, inf), but {self.alpha} was given!")

    def n_samples(self):
        """ceil(exp(coding_bits * log 2)) in float32 (importance_sampling.py:50)."""
        return int(_lib.load().irec_importance_n_samples(float(self.coding_bits)))

    def coded_sample(self, target, coder, seed):
        """samplers.py:73-83 -> encode_gaussian_importance_sample: (index, sample); `sample` has coder.loc's shape,
        dtype float32, on coder.loc's device."""
        self._check_alpha()
        loc = torch.as_tensor(coder.loc)
        tl, ts, pl, ps = (_host_f32(t) for t in (target.loc, target.scale, coder.loc, coder.scale))
        if not (tl.size == ts.size == pl.size == ps.size):
            raise CodingError("target and coder must have the same shape")
        out = np.empty_like(pl)
        idx = ctypes.c_int64(-1)
        _lib.check(_lib.load().irec_importance_encode(_ptr(tl), _ptr(ts), _ptr(pl), _ptr(ps), pl.size,
                                                      float(self.coding_bits), float(self.alpha), int(seed),
                                                      ctypes.byref(idx), _ptr(out)),
                   "irec_importance_encode")
        return int(idx.value), torch.from_numpy(out).reshape(loc.shape).to(loc.device)

    def decode_sample(self, coder, sample_index, seed):
        """samplers.py:85-92 -> decode_gaussian_importance_sample."""
        loc = torch.as_tensor(coder.loc)
        pl, ps = _host_f32(coder.loc), _host_f32(coder.scale)
        out = np.empty_like(pl)
        _lib.check(_lib.load().irec_importance_decode(_ptr(pl), _ptr(ps), pl.size, int(sample_index), int(seed), _ptr(out)),
                   "irec_importance_decode")
        return torch.from_numpy(out).reshape(loc.shape).to(loc.device)

    def update(self, target, coder):
        print("ImportanceSampler doesn't require updating!")   # samplers.py:94-97

    def get_codelength(self, index):
        """samplers.py:99-100: coding_bits * log 2 nats, whatever the index."""
        return float(np.float32(self.coding_bits) * np.float32(math.log(2.)))
