"""Host mirror of rec/coding/beam_search_coder.py: same constructor, methods, argument order and error behaviour;
the arithmetic runs in the gfx950 kernels behind include/irec.h (no CPU path).

Distributions are duck-typed exactly as in the reference (only `.loc` and `.scale` are read, coder.py:427-430),
e.g. torch.distributions.Normal with CPU or cuda (HIP) float32 tensors.
"""
import numpy as np
import torch

from .coder import GaussianCoder
from .utils import CodingError
from .. import _lib
from ..engine import get_engine


class BeamSearchCoder(GaussianCoder):
    def __init__(self, kl_per_partition, n_beams, extra_samples=1., extrapolate_auxiliary_ratios=True,
                 name="gaussian_encoder", **kwargs):
        """beam_search_coder.py:15-30."""
        super().__init__(name=name, kl_per_partition=kl_per_partition, sampler=None,
                         extrapolate_auxiliary_ratios=extrapolate_auxiliary_ratios, **kwargs)
        self.n_beams = n_beams
        self.n_samples = int(np.exp(kl_per_partition * extra_samples))
        self.big_prime = 10007
        self.force_generic = False   # debugging / testing knob: IREC_FLAG_FORCE_GENERIC
        self.fused_philox = False    # debugging / testing knob: IREC_FLAG_FUSED_PHILOX
        self.one_table = False       # debugging / testing knob: IREC_FLAG_ONE_TABLE
        self.team = False            # debugging / testing knob: IREC_FLAG_TEAM (the team encoder also for small calls)
        self._max_K_hint = 32

    # ---- small host-side mirrors ---------------------------------------------------------------------------------
    def simple_hash(self, matrix):
        """beam_search_coder.py:33-35 (int32 arithmetic, floormod)."""
        m = np.asarray(matrix, dtype=np.int64).reshape(len(matrix), -1)
        w = np.arange(69, 69 + m.shape[1], dtype=np.int64)
        s = ((m * w).sum(axis=1) + 2 ** 31) % 2 ** 32 - 2 ** 31  # int32 wrap-around
        return (np.mod(s, self.big_prime - 1) + 1).astype(np.int32)

    def get_codelength(self, indicies):
        """beam_search_coder.py:150-151."""
        return len(indicies) * np.log(self.n_samples)

    # ---- plumbing --------------------------------------------------------------------------------------------------
    def _params(self):
        if not self.extrapolate_auxiliary_ratios:
            raise CodingError("only extrapolate_auxiliary_ratios=True is supported on the beam-search path")
        if not (1 <= self.n_beams <= _lib.MAX_BEAMS):
            raise CodingError(f"n_beams must be in [1, {_lib.MAX_BEAMS}], got {self.n_beams}")
        if self.n_samples < 1:
            raise CodingError(f"n_samples = {self.n_samples} < 1")
        flags = (_lib.IREC_FLAG_FORCE_GENERIC if self.force_generic else 0) | \
                (_lib.IREC_FLAG_FUSED_PHILOX if self.fused_philox else 0) | \
                (_lib.IREC_FLAG_ONE_TABLE if self.one_table else 0) | \
                (_lib.IREC_FLAG_TEAM if self.team else 0)
        return get_engine().params(self.kl_per_partition, self.n_samples, self.n_beams, flags)

    @staticmethod
    def _dev(t, device):
        t = torch.as_tensor(t)
        return t.detach().to(device=device, dtype=torch.float32).contiguous()

    def _engine_for(self, tensor):
        t = torch.as_tensor(tensor)
        return get_engine(t.device if t.device.type == "cuda" else None)

    def encode_tensors(self, q_loc, q_scale, p_loc, p_scale, seed, block_size):
        """Batched core of encode / encode_block: leading dim = independent latent tensors.
        Returns (indices per tensor per block, sample tensor on the input's device)."""
        src = torch.as_tensor(q_loc)
        eng = self._engine_for(src)
        params = self._params()
        n_tensors = src.shape[0]
        n = src[0].numel()
        shapes = {tuple(torch.as_tensor(t).shape) for t in (q_loc, q_scale, p_loc, p_scale)}
        if len(shapes) != 1:
            raise CodingError("All tensor arguments supplied to split must have the same batch dimensions!")
        ql, qs, pl, ps = (self._dev(t, eng.device) for t in (q_loc, q_scale, p_loc, p_scale))
        lay = eng.layout(n_tensors, n, block_size, seed)
        max_K = self._max_K_hint
        while True:
            K, idx, sample = eng.encode_blocks(params, lay, ql, qs, pl, ps, seed, max_K)
            K_host = K.cpu().numpy()
            if (K_host < 0).any():
                raise CodingError("a block exceeded the engine's dimension bound")
            need = int(K_host.max()) if K_host.size else 0
            if need <= max_K:
                break
            if need > _lib.MAX_PARTITIONS:
                raise CodingError(f"KL divergence needs {need} partitions; this build supports {_lib.MAX_PARTITIONS}")
            max_K = need
            self._max_K_hint = max(self._max_K_hint, need)
        idx_host = idx.cpu().numpy()
        per_tensor = []
        bpt = lay.blocks_per_tensor
        for i in range(n_tensors):
            blocks = []
            for j in range(bpt):
                row = lay.natural[i * bpt + j]
                blocks.append([int(v) for v in idx_host[row, :K_host[row]]])
            per_tensor.append(blocks)
        return per_tensor, sample.reshape(src.shape).to(src.device)

    def decode_tensors(self, p_loc, p_scale, indices, seed, block_size):
        src = torch.as_tensor(p_loc)
        eng = self._engine_for(src)
        params = self._params()
        n_tensors = src.shape[0]
        n = src[0].numel()
        pl, ps = (self._dev(t, eng.device) for t in (p_loc, p_scale))
        lay = eng.layout(n_tensors, n, block_size, seed)
        bpt = lay.blocks_per_tensor
        if len(indices) != n_tensors or any(len(b) != bpt for b in indices):
            raise CodingError("indices do not match the block structure of coding_dist")
        max_K = max(1, max((len(ix) for b in indices for ix in b), default=1))
        K = np.zeros(lay.n_blocks, dtype=np.int32)
        idx = np.zeros((lay.n_blocks, max_K), dtype=np.int32)
        for i in range(n_tensors):
            for j in range(bpt):
                row = lay.natural[i * bpt + j]
                ix = indices[i][j]
                K[row] = len(ix)
                idx[row, :len(ix)] = np.asarray(ix, dtype=np.int32)
        if idx.size and (idx.min() < 0 or idx.max() >= self.n_samples):
            raise CodingError("index out of range [0, n_samples)")
        sample = eng.decode_blocks(params, lay, pl, ps, seed, torch.from_numpy(K).to(eng.device),
                                   torch.from_numpy(idx).to(eng.device))
        return sample.reshape(src.shape).to(src.device)

    # ---- reference API -----------------------------------------------------------------------------------------------
    def encode_block(self, target_dist, coding_dist, seed, update_sampler=False, numpy=True):
        """beam_search_coder.py:53-122.  Returns (list of K indices, sample with the shape of loc)."""
        if target_dist.loc.shape[0] != 1:
            raise CodingError("For encoding, batch size must be 1.")
        idx, sample = self.encode_tensors(target_dist.loc, target_dist.scale, coding_dist.loc, coding_dist.scale,
                                          seed, None)
        indices = idx[0][0]
        if numpy:
            indices = [np.int32(v) for v in indices]
        return list(indices), sample

    def decode_block(self, coding_dist, indices, seed):
        """beam_search_coder.py:124-148.  (The reference reverses the caller's list in place, :127; this does not.)"""
        indices = [int(v) for v in indices]
        loc = torch.as_tensor(coding_dist.loc)
        batched = loc if loc.shape[0] == 1 else loc.reshape((1,) + tuple(loc.shape))
        scale = torch.as_tensor(coding_dist.scale).reshape(batched.shape)
        out = self.decode_tensors(batched, scale, [[indices]], seed, None)
        return out.reshape(loc.shape)

    def encode(self, target_dist, coding_dist, seed, **kwargs):
        """GaussianCoder.encode, coder.py:412-457.  Extension: a leading batch dim N > 1 encodes N independent latent
        tensors in one launch when `batched=True` is passed (the reference rejects N != 1, beam_search_coder.py:54-55)."""
        batched = kwargs.pop("batched", False)
        if self.block_size is None and not batched:
            return self.encode_block(target_dist, coding_dist, seed, **kwargs)
        if target_dist.loc.shape[0] != 1 and not batched:
            raise CodingError("For encoding, batch size must be 1.")
        idx, sample = self.encode_tensors(target_dist.loc, target_dist.scale, coding_dist.loc, coding_dist.scale,
                                          seed, self.block_size)
        if self.block_size is None:
            idx = [blocks[0] for blocks in idx]
        return (idx if batched else idx[0]), sample

    def decode(self, coding_dist, indices, seed, **kwargs):
        """GaussianCoder.decode, coder.py:459-491."""
        batched = kwargs.pop("batched", False)
        if self.block_size is None and not batched:
            return self.decode_block(coding_dist, indices, seed)
        per_tensor = indices if batched else [indices]
        if self.block_size is None:
            per_tensor = [[ix] for ix in per_tensor]
        return self.decode_tensors(coding_dist.loc, coding_dist.scale, per_tensor, seed, self.block_size)
