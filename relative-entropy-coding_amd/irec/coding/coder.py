"""Host mirror of rec/coding/coder.py for the beam-search path (reference file:line in each docstring).

Only what `sampler='beam_search'` needs is implemented: the block split/merge bookkeeping and the extrapolated
auxiliary-variance ratios.  The sampler-driven GaussianCoder.encode_block (coder.py:493-584) and the SGD ratio fitter
(coder.py:233-410) are out of scope.
"""
import abc

import numpy as np
import torch

from .utils import CodingError
from ..engine import tf_shuffle_perm

AUX_RATIO_POWER_LAW = -0.7864636765648174  # coder.py:16


class Coder(abc.ABC):
    """coder.py:27-138.  `split`/`merge` are kept for API parity and for host-side checks; the kernels perform the
    same gather / scatter through the permutation on the fly."""

    def __init__(self, block_size=None, name="encoder", **kwargs):
        self.name = name
        self.block_size = block_size

    def split(self, *args, seed=42):
        """coder.py:38-85: flatten, shuffle all tensors with the same seeded permutation, cut into blocks."""
        tensor_shape = args[0].shape
        flattened = []
        for tensor in args:
            if tensor.shape != tensor_shape:
                raise CodingError("All tensor arguments supplied to split must have the same batch dimensions!")
            flattened.append(tensor.reshape(-1))
        num_dims = flattened[0].shape[0]
        perm = torch.from_numpy(tf_shuffle_perm(seed, num_dims)).to(flattened[0].device)
        flattened = [flat[perm] for flat in flattened]
        all_blocks = []
        for tensor in flattened:
            all_blocks.append([tensor[i:min(i + self.block_size, num_dims)]
                               for i in range(0, num_dims, self.block_size)])
        return all_blocks

    def merge(self, *args, shape=None, seed=42):
        """coder.py:87-122: inverse of split."""
        if shape is None:
            raise CodingError("Shape cannot be None!")
        tensors = [torch.cat(list(blocks), dim=0) for blocks in args]
        num_dims = tensors[0].shape[0]
        for tensor in tensors:
            if tensor.dim() != 1:
                raise CodingError("All supplied tensors to merge must be rank 1!")
            if tensor.shape[0] != num_dims:
                raise CodingError("All tensors must have the same number of dimensions!")
        perm = torch.from_numpy(tf_shuffle_perm(seed, num_dims)).to(tensors[0].device)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(num_dims, device=perm.device)
        return [tensor[inv].reshape(shape) for tensor in tensors]

    @abc.abstractmethod
    def encode(self, target_dist, coding_dist, seed, **kwargs):
        pass

    @abc.abstractmethod
    def decode(self, coding_dist, indices, seed, **kwargs):
        pass

    @abc.abstractmethod
    def encode_block(self, target_dist, coding_dist, seed, **kwargs):
        pass

    @abc.abstractmethod
    def decode_block(self, coding_dist, indices, seed, **kwargs):
        pass


class GaussianCoder(Coder):
    """coder.py:174-231 (constructor + get_auxiliary_ratio)."""

    def __init__(self, kl_per_partition, sampler=None, extrapolate_auxiliary_ratios=True, block_size=None,
                 name="gaussian_encoder", **kwargs):
        super().__init__(name=name, block_size=block_size, **kwargs)
        self.sampler = sampler
        self.kl_per_partition = np.float32(kl_per_partition)  # tf.cast(kl_per_partition, tf.float32), coder.py:192
        self.extrapolate_auxiliary_ratios = extrapolate_auxiliary_ratios

    def get_auxiliary_ratio(self, index):
        """coder.py:218-220 (extrapolated power law only)."""
        if self.extrapolate_auxiliary_ratios:
            return np.power(index + 1., AUX_RATIO_POWER_LAW)
        raise CodingError("Coder has not been initialized yet, please use extrapolation: fitted auxiliary "
                          "variance ratios (update_auxiliary_variance_ratios) are outside the beam-search path")

    def update_auxiliary_variance_ratios(self, target_dist, coding_dist, seed=42, **kwargs):
        """coder.py:233-264.  A no-op with extrapolated ratios (the coder is stateless)."""
        if not self.extrapolate_auxiliary_ratios:
            raise CodingError("fitting auxiliary variance ratios is outside the beam-search path")

    def encode(self, target_dist, coding_dist, seed, **kwargs):
        raise CodingError("GaussianCoder with a rejection/importance sampler is outside the beam-search path; "
                          "use BeamSearchCoder")

    def decode(self, coding_dist, indices, seed, **kwargs):
        raise CodingError("GaussianCoder with a rejection/importance sampler is outside the beam-search path; "
                          "use BeamSearchCoder")

    def encode_block(self, target_dist, coding_dist, seed, **kwargs):
        raise CodingError("GaussianCoder.encode_block (sequential sampler) is outside the beam-search path")

    def decode_block(self, coding_dist, indices, seed, **kwargs):
        raise CodingError("GaussianCoder.decode_block (sequential sampler) is outside the beam-search path")
