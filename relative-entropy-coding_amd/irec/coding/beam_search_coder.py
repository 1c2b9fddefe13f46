"""Host mirror of rec/coding/beam_search_coder.py: same constructor, methods, argument order and error behaviour;
the arithmetic runs in the gfx950 kernels behind include/irec.h (no CPU path).

Distributions are duck-typed exactly as in the reference (only `.loc` and `.scale` are read, coder.py:427-430),
e.g. torch.distributions.Normal with CPU or cuda (HIP) float32 tensors.
"""
import gc

import numpy as np
import torch

from .coder import GaussianCoder
from .utils import CodingError
from .. import _lib
from ..engine import Engine, get_engine


class MorePartitionsNeeded(CodingError):
    """A block has K = ceil(KL / Omega) > max_K: it was NOT coded; encode again with max_K >= need."""

    def __init__(self, need):
        super().__init__(f"a block needs {need} partitions: encode again with max_K >= {need}")
        self.need = need


class SplitNotResident(CodingError):
    """A cooperative encoder (the split encoder: several workgroups per block of a small call; shared rows: several teams per
    row of a mid-size call) gave up waiting for partners that were not resident -- other work held the CUs.  The blocks were NOT
    coded.  The coder that issued the call has been told (`BeamSearchCoder._split_gave_up`): its next call -- the one that
    codes these blocks again -- goes out without sharing; see there for when sharing comes back."""


class PendingCode:
    """Result of one asynchronous encode call: everything stays on the device until the host asks.
    K [n_blocks] int32, idx [n_blocks, max_K] int32 (rows in layout order), sample (input shape)."""

    def __init__(self, coder, lay, K, idx, sample, max_K, shared=False, params=None):
        self.coder, self.lay, self.K, self.idx, self.sample, self.max_K = coder, lay, K, idx, sample, max_K
        self.shared, self.params = shared, params   # the call was allowed to share blocks between workgroups / teams; its irec_params

    def _check(self, K_host):
        """Errors and hints from the partition counts read back (shared by the list and the packed read-backs)."""
        if (K_host == -2).any():
            raise SplitNotResident("the cooperating workgroups of a shared block were not all resident (other work on this device, "
                                   "or two cooperating calls in flight on different streams): the call is coded again without sharing")
        if self.shared and self.coder._split_strikes and self.params is not None and K_host.size:
            # a call that did share blocks came back whole: the device is ours again, the back-off starts over
            eng = self.coder._engine_for(self.K)
            if eng.plan(self.params, self.lay, self.max_K)["split"] >= 2:
                self.coder._split_strikes = 0
        if (K_host < 0).any():
            raise CodingError("a block exceeded the engine's dimension bound")
        need = int(K_host.max()) if K_host.size else 0
        limit = getattr(self.coder._engine_for(self.K), "max_partitions", _lib.MAX_PARTITIONS) if not self.coder.extrapolate_auxiliary_ratios else _lib.MAX_PARTITIONS
        if need > limit:                 # fitted ratios: GaussianCoder.get_auxiliary_ratio raises for the first index past the table
            self.coder.get_auxiliary_ratio(need - 1)
        if need > _lib.MAX_PARTITIONS:   # (before any hint is touched: one degenerate block -- an infinite KL reads back as
            # 10^9 partitions -- must not leave the coder, or the model that shares its hints, asking for more than the library takes)
            raise CodingError(f"KL divergence needs {need} partitions; this build supports {_lib.MAX_PARTITIONS}")
        self.coder._max_K_hint = min(max(self.coder._max_K_hint, need), _lib.MAX_PARTITIONS)
        c = self.coder            # table window hint: what covers the bulk of the blocks read back (a few outliers take the
        if K_host.size:           # second pass instead of stretching every later call's tables), decaying by an eighth per read
            flat = K_host.reshape(-1)
            upper_quartile = int(np.partition(flat, (3 * (flat.size - 1)) // 4)[(3 * (flat.size - 1)) // 4])   # (np.quantile
            bulk = (5 * upper_quartile + 3) // 4                       #  costs 40 us a call: 1 ms per 24-block image)
            c._K_seen = bulk if c._K_reads == 0 else max(bulk, c._K_seen - max(1, c._K_seen // 8))
        c._K_reads += 1
        if need > self.max_K:
            raise MorePartitionsNeeded(need)

    def _lists(self, K_host, idx_host):
        self._check(K_host)
        lay = self.lay
        bpt = lay.blocks_per_tensor
        # One C-speed conversion, the per-block lists are slices of it; the cyclic collector is paused meanwhile (a batched
        # model pass makes tens of thousands of small lists, none of them garbage: generation-2 passes triggered by the
        # allocation count alone cost more than the conversion).
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            rows, ks = idx_host.tolist(), K_host.tolist()
            nat = lay.natural.tolist() if hasattr(lay.natural, "tolist") else list(lay.natural)
            return [[rows[r][:ks[r]] for r in nat[i * bpt:(i + 1) * bpt]] for i in range(lay.n_tensors)]
        finally:
            if gc_was_on:
                gc.enable()

    def to_lists(self):
        """Indices per tensor per block (host lists): ONE device-to-host copy (K and the index rows together)."""
        both = torch.cat([self.K[:, None], self.idx], dim=1).cpu().numpy()
        return self._lists(both[:, 0], both[:, 1:])

    @staticmethod
    def gather_packed(pendings):
        """Indices of R calls on the SAME layout (the residual blocks of a batched model pass) as packed arrays, image-major:
        K [N, R, bpt] int32, idx [N, R, bpt, max_K] int32 (numpy; rows are valid up to K) -- ONE gather on the device and ONE
        device-to-host copy, no per-index Python object.  irec.io.encode_files takes them as they are."""
        lay = pendings[0].lay
        same = all((p.lay.n_tensors, p.lay.n, p.lay.block_size, p.lay.n_blocks) == (lay.n_tensors, lay.n, lay.block_size, lay.n_blocks)
                   and (p.lay.block_size is None or p.lay.seed == lay.seed) for p in pendings)
        if not same or lay.natural is None:
            raise CodingError("gather_packed needs calls on one block layout (same tensor count, size, block_size and seed)")
        width = max(p.idx.shape[1] for p in pendings)
        rows = []
        for p in pendings:
            r = torch.cat([p.K[:, None], p.idx], dim=1)
            if r.shape[1] < width + 1:
                r = torch.nn.functional.pad(r, (0, width + 1 - r.shape[1]))
            rows.append(r)
        n, bpt, R = lay.n_tensors, lay.blocks_per_tensor, len(pendings)
        sel = lay.packed_index(R)                                             # row of (image i, residual block r, block j)
        both = torch.cat(rows, dim=0).index_select(0, sel).cpu().numpy().reshape(n, R, bpt, width + 1)
        K, idx = both[..., 0], both[..., 1:]
        retry, split_failed = None, False
        for r, p in enumerate(pendings):
            try:
                p._check(K[:, r, :])
            except MorePartitionsNeeded as e:
                retry = e if retry is None or not isinstance(retry, MorePartitionsNeeded) or e.need > retry.need else retry
            except SplitNotResident as e:
                split_failed = True
                retry = retry if isinstance(retry, MorePartitionsNeeded) else e
        if split_failed:
            PendingCode._all_gave_up(pendings)
        if retry is not None:
            raise retry
        return np.ascontiguousarray(K), np.ascontiguousarray(idx)

    @staticmethod
    def _all_gave_up(pendings):
        """One call of a pass (the residual blocks of an image) gave up: every coder of the pass steps back, once."""
        seen = set()
        for p in pendings:
            if id(p.coder) not in seen:
                seen.add(id(p.coder))
                p.coder._split_gave_up()

    @staticmethod
    def gather(pendings):
        """Indices of many calls (e.g. the 24 residual blocks of a model pass) with ONE device-to-host copy in all."""
        if not pendings:
            return []
        width = max(p.idx.shape[1] for p in pendings)
        rows = []
        for p in pendings:
            r = torch.cat([p.K[:, None], p.idx], dim=1)
            if r.shape[1] < width + 1:
                r = torch.nn.functional.pad(r, (0, width + 1 - r.shape[1]))
            rows.append(r)
        both = torch.cat(rows, dim=0).cpu().numpy()
        out, at, retry, split_failed = [], 0, None, False
        for p in pendings:
            n = p.K.shape[0]
            try:
                out.append(p._lists(both[at:at + n, 0], both[at:at + n, 1:1 + p.idx.shape[1]]))
            except MorePartitionsNeeded as e:     # keep going: every coder's hint is raised before the caller codes again
                retry = e if retry is None or e.need > retry.need else retry
            except SplitNotResident as e:         # likewise: every coder of the pass steps back before it is coded again
                retry = retry if isinstance(retry, MorePartitionsNeeded) else e
                split_failed = True
            at += n
        if split_failed:
            PendingCode._all_gave_up(pendings)
        if retry is not None:
            raise retry
        return out


class BeamSearchCoder(GaussianCoder):
    def __init__(self, kl_per_partition, n_beams, extra_samples=1., extrapolate_auxiliary_ratios=True,
                 name="gaussian_encoder", **kwargs):
        """beam_search_coder.py:15-30.  Extension: `engine=` (an `irec.Engine` built with the caller's own quantile table,
        irec_create_ex) instead of the device's default engine."""
        self.engine = kwargs.pop("engine", None)
        super().__init__(name=name, kl_per_partition=kl_per_partition, sampler=None,
                         extrapolate_auxiliary_ratios=extrapolate_auxiliary_ratios, **kwargs)
        self.n_beams = n_beams
        self.n_samples = int(np.exp(kl_per_partition * extra_samples))
        self.big_prime = 10007
        self.force_generic = False   # debugging / testing knob: IREC_FLAG_FORCE_GENERIC
        self.fused_philox = False    # debugging / testing knob: IREC_FLAG_FUSED_PHILOX
        self.one_table = False       # debugging / testing knob: IREC_FLAG_ONE_TABLE
        self.team = False            # debugging / testing knob: IREC_FLAG_TEAM (the team encoder also for small calls)
        self.no_split = False        # the caller's knob: IREC_FLAG_NO_SPLIT on every call (no block is ever shared between workgroups / teams)
        self._split_strikes = 0      # consecutive give-ups of the cooperative encoders (SplitNotResident), see _split_gave_up
        self._split_pause = 0        # calls still to be issued without sharing
        self._test_split_orphan = False  # test hook (IREC_FLAG_TEST_SPLIT_ORPHAN): the split encoder's partners leave at once
        self.team_shape = "default"  # diagnostics: IREC_FLAG_SHAPE_* workgroup shape of the team encoder
        self.reuse_tables = True     # IREC_FLAG_REUSE_TABLES: a call whose proposal tables are already in the stream's scratch
                                     # (same seed, S, dims and window: the 24 residual blocks of an image) does not rebuild them
        self.table_steps = 0         # partitions the per-call proposal tables cover; 0 = sized from the partition counts
                                     # this coder has seen so far (_K_seen + 4, at least 8; the library bounds the bytes);
                                     # blocks with more are coded by the fused-Philox kernel in a second pass of the same call
        self._max_K_hint = 32
        self._K_seen = 28            # largest K read back so far (starts at the default window; shrinks with evidence)
        self._K_reads = 0

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
    def table_window(self):
        """Partitions the per-call proposal tables cover: `table_steps`, or the running hint from the partition counts read
        back so far (the tables cost set-up time and scratch per step they cover; typical K is ~8)."""
        if self.table_steps:
            return int(self.table_steps)
        return min(_lib.IREC_TABLE_STEPS_MAX, max(8, (self._K_seen + 4 + 3) // 4 * 4))

    # ---- recovery from a give-up (round 5; until then one give-up set no_split for the life of the coder) -----------
    SPLIT_PAUSE_MAX = 64
    SPLIT_STRIKES_FINAL = 8    # consecutive give-ups after which the coder never shares a block again (no_split = True)

    def _split_gave_up(self):
        """A cooperative call of this coder read back K = -2 (partners not resident within 100 ms).  Bounded back-off: the
        n-th consecutive give-up keeps the next 2^(n-1) calls -- the first of which codes the failed blocks again -- free of
        shared blocks (1, 2, 4, ... at most SPLIT_PAUSE_MAX calls); then the cooperative form is tried again, and the first
        shared call that comes back whole resets the count.  One transient co-tenant therefore costs ONE call its 100 ms and
        a recode, not every later small call a factor of two."""
        self._split_strikes = min(self._split_strikes + 1, 1 + self.SPLIT_PAUSE_MAX.bit_length())
        self._split_pause = min(1 << (self._split_strikes - 1), self.SPLIT_PAUSE_MAX)
        if self._split_strikes >= self.SPLIT_STRIKES_FINAL:
            # a PERMANENT co-tenant (another stream's gangs hold the CUs for good): every retry costs its 100 ms, a recode and -- under
            # HIP-graph replay -- a re-capture; after this many give-ups in a row (127 calls of back-off in between) the coder stops sharing
            self.no_split = True

    def _take_sharing_turn(self):
        """May the call about to be issued share blocks?  Consumes one call of a running pause."""
        if self.no_split:
            return False
        if self._split_pause > 0:
            self._split_pause -= 1
            return False
        return True

    def _params(self, table_steps=None, shared=None):
        if shared is None:
            shared = not self.no_split and self._split_pause == 0
        if not self.extrapolate_auxiliary_ratios:
            self.get_auxiliary_ratio(0)          # (raises the reference's "has not been initialized yet", coder.py:222-225)
        if not (1 <= self.n_beams <= _lib.MAX_BEAMS):
            raise CodingError(f"n_beams must be in [1, {_lib.MAX_BEAMS}], got {self.n_beams}")
        if self.n_samples < 1:
            raise CodingError(f"n_samples = {self.n_samples} < 1")
        flags = (_lib.IREC_FLAG_FORCE_GENERIC if self.force_generic else 0) | \
                (_lib.IREC_FLAG_FUSED_PHILOX if self.fused_philox else 0) | \
                (_lib.IREC_FLAG_ONE_TABLE if self.one_table else 0) | \
                (_lib.IREC_FLAG_TEAM if self.team else 0) | (0 if shared else _lib.IREC_FLAG_NO_SPLIT) | \
                (_lib.IREC_FLAG_REUSE_TABLES if self.reuse_tables else 0) | \
                _lib.IREC_FLAG_SHAPE[self.team_shape] | \
                (_lib.IREC_FLAG_TEST_SPLIT_ORPHAN if self._test_split_orphan and shared else 0)
        steps = int(table_steps) if table_steps else self.table_window()
        return Engine.params(self.kl_per_partition, self.n_samples, self.n_beams, flags, table_steps=steps)

    @staticmethod
    def _dev(t, device):
        t = torch.as_tensor(t)
        return t.detach().to(device=device, dtype=torch.float32).contiguous()

    def _engine_for(self, tensor):
        if self.engine is not None and self.extrapolate_auxiliary_ratios:
            return self.engine
        t = torch.as_tensor(tensor)
        if self.extrapolate_auxiliary_ratios:
            return get_engine(t.device if t.device.type == "cuda" else None)
        # fitted ratios (extrapolate_auxiliary_ratios=False, coder.py:203-231): a context of this coder's own carries them
        # The context is keyed on what it was built from -- the device and the ratio table's bytes: assigning
        # aux_variable_variance_ratios directly (the analogue of the reference restoring its tf.Variables from a checkpoint) or
        # coding a tensor that lives on another GPU builds a new one instead of coding on a stale table.
        self.get_auxiliary_ratio(0)
        dev = t.device if t.device.type == "cuda" else (self.engine.device if self.engine is not None else
                                                        torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None)
        if dev is None:
            raise _lib.IrecLibraryError("irec needs a HIP device (MI355X / gfx950); there is no CPU fallback")
        ratios = np.ascontiguousarray(np.asarray(self.aux_variable_variance_ratios, dtype=np.float32))
        key = (str(torch.device(dev)), ratios.tobytes())
        if getattr(self, "_ratio_engine", None) is None or getattr(self, "_ratio_engine_key", None) != key:
            self._ratio_engine = Engine(dev, lut=getattr(self.engine, "lut", None), aux_ratios=ratios)
            self._ratio_engine_key = key
        return self._ratio_engine

    def encode_tensors_device(self, q_loc, q_scale, p_loc, p_scale, seed, block_size, max_K=None, table_steps=None):
        """Asynchronous core of encode: launches the encoder on the current stream and returns a `PendingCode` -- K,
        indices and sample still on the device, NO host synchronisation.  The caller reads the indices when it needs
        them (`PendingCode.to_lists`, one device-to-host copy; `PendingCode.gather` for many calls at once)."""
        src = torch.as_tensor(q_loc)
        if src.ndim < 2 or src.shape[0] < 1 or src[0].numel() < 1:
            raise CodingError(f"nothing to encode: distributions of shape {tuple(src.shape)} (need [batch >= 1, dims >= 1 ...])")
        eng = self._engine_for(src)
        shared = self._take_sharing_turn()
        params = self._params(table_steps, shared)   # (a model passes ONE window to all its coders: equal keys, tables built once)
        n_tensors = src.shape[0]
        n = src[0].numel()
        shapes = {tuple(torch.as_tensor(t).shape) for t in (q_loc, q_scale, p_loc, p_scale)}
        if len(shapes) != 1:
            raise CodingError("All tensor arguments supplied to split must have the same batch dimensions!")
        ql, qs, pl, ps = (self._dev(t, eng.device) for t in (q_loc, q_scale, p_loc, p_scale))
        lay = eng.layout(n_tensors, n, block_size, seed)
        max_K = self._max_K_hint if max_K is None else int(max_K)
        K, idx, sample = eng.encode_blocks(params, lay, ql, qs, pl, ps, seed, max_K)
        return PendingCode(self, lay, K, idx, sample.reshape(src.shape).to(src.device), max_K, shared, params)

    def encode_tensors(self, q_loc, q_scale, p_loc, p_scale, seed, block_size):
        """Batched core of encode / encode_block: leading dim = independent latent tensors.
        Returns (indices per tensor per block, sample tensor on the input's device)."""
        max_K = None
        while True:
            pending = self.encode_tensors_device(q_loc, q_scale, p_loc, p_scale, seed, block_size, max_K)
            try:
                return pending.to_lists(), pending.sample
            except MorePartitionsNeeded as e:     # a block's KL asks for more partitions than the index buffer holds
                max_K = e.need
            except SplitNotResident:              # partners not resident (the device is shared): the next call -- this loop's --
                if not pending.shared:            # codes the blocks again without sharing, which cannot wait in vain
                    raise
                self._split_gave_up()
    def decode_tensors(self, p_loc, p_scale, indices, seed, block_size):
        src = torch.as_tensor(p_loc)
        if src.ndim < 2 or src.shape[0] < 1 or src[0].numel() < 1:
            raise CodingError(f"nothing to decode: coding distribution of shape {tuple(src.shape)} (need [batch >= 1, dims >= 1 ...])")
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
        if not self.extrapolate_auxiliary_ratios:
            # decode_block asks get_auxiliary_ratio(i) for i = len(indices) - 1 .. 0 (beam_search_coder.py:129-131): an index list longer
            # than the fitted table raises the reference's error here -- the kernels would only mark such a row "not decodable" (p.loc)
            self.get_auxiliary_ratio(max_K - 1)
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
        """GaussianCoder.encode, coder.py:412-457.  Extensions (the reference rejects N != 1, beam_search_coder.py:54-55):
        `batched=True`: a leading batch dim N > 1 encodes N independent latent tensors in one launch;
        `defer=True`: returns (PendingCode, sample) without any host synchronisation -- the model shims use it to keep
        the 24 sequential residual blocks of an image on the device and read all indices once at the end."""
        batched = kwargs.pop("batched", False)
        defer = kwargs.pop("defer", False)
        if self.block_size is None and not batched and not defer:
            return self.encode_block(target_dist, coding_dist, seed, **kwargs)
        if target_dist.loc.shape[0] != 1 and not batched:
            raise CodingError("For encoding, batch size must be 1.")
        if defer:
            pending = self.encode_tensors_device(target_dist.loc, target_dist.scale, coding_dist.loc, coding_dist.scale,
                                                 seed, self.block_size, kwargs.pop("max_K", None),
                                                 kwargs.pop("table_steps", None))
            return pending, pending.sample
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
