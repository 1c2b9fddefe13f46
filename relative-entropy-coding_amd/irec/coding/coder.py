"""Host mirror of rec/coding/coder.py for the beam-search path (reference file:line in each docstring).

Only what `sampler='beam_search'` needs is implemented: the block split/merge bookkeeping and the auxiliary-variance
ratios -- extrapolated (the power law), or FITTED ones handed over as data (round 5).  The sampler-driven
GaussianCoder.encode_block (coder.py:493-584) and the SGD ratio fitter that produces fitted ratios (coder.py:233-410) are
out of scope.
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
        if not self.extrapolate_auxiliary_ratios:          # coder.py:203-216: the variables a checkpoint restores
            self.aux_variable_variance_ratios = np.array([1.], dtype=np.float32)
            self._initialized = False

    def set_auxiliary_variance_ratios(self, ratios):
        """The FITTED ratios of an extrapolate_auxiliary_ratios=False coder, as data: what the reference restores into
        `aux_variable_variance_ratios` / `_initialized` from a checkpoint (coder.py:203-216).  The fitter itself
        (update_auxiliary_variance_ratios, coder.py:233-410) stays on the caller's side."""
        if self.extrapolate_auxiliary_ratios:
            raise CodingError("this coder extrapolates its auxiliary ratios (extrapolate_auxiliary_ratios=True)")
        r = np.ascontiguousarray(np.asarray(ratios, dtype=np.float32).reshape(-1))
        if r.size < 1 or not np.all((r > 0) & (r <= 1)):
            raise CodingError("auxiliary variance ratios must be a non-empty sequence of numbers in (0, 1]")
        self.aux_variable_variance_ratios = r
        self._initialized = True
        self._ratio_engine = None

    def get_auxiliary_ratio(self, index):
        """coder.py:218-231."""
        if self.extrapolate_auxiliary_ratios:
            return np.power(index + 1., AUX_RATIO_POWER_LAW)
        if not self._initialized:
            raise CodingError("Coder has not been initialized yet, please call"
                              "update_auxiliary_variance_ratios() first"
                              " or use extrapolation")
        if index >= self.aux_variable_variance_ratios.shape[0]:
            raise CodingError("KL divergence higher than auxiliary variables can account for. "
                              "Update auxiliary variable ratios with high-enough KL divergence."
                              "Maximum possible number of partitions is {}."
                              "Requested {}".format(self.aux_variable_variance_ratios.shape[0], index + 1))
        return self.aux_variable_variance_ratios[index]

    def update_auxiliary_variance_ratios(self, target_dist, coding_dist, seed=42, **kwargs):
        """coder.py:233-264.  A no-op with extrapolated ratios (the coder is stateless); the SGD fit of
        coder.py:265-410 is out of scope -- hand fitted ratios over with set_auxiliary_variance_ratios."""
        if not self.extrapolate_auxiliary_ratios:
            raise CodingError("fitting auxiliary variance ratios is outside the beam-search path: "
                              "set_auxiliary_variance_ratios(ratios) takes fitted ones as data")

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
