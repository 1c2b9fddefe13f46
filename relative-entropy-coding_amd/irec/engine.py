"""Per-device engine: owns the irec_context, the cached block descriptors / permutations / scratch slabs, and
launches the C-ABI entry points on the current torch stream.  PyTorch is used for device memory and streams only.

Host-side counterpart of the loops in the reference's
    GaussianCoder.encode / decode   rec/coding/coder.py:412-491
    Coder.split / merge             rec/coding/coder.py:38-122   (here: gather/scatter through `perm` inside the kernels)
"""
import ctypes
import threading

import numpy as np
import torch

from . import _lib

_engines = {}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def tf_shuffle_perm(seed, n):
    """tf.random.set_seed(seed); tf.random.shuffle(tf.range(n))  (reference: coder.py:62-64).  Host numpy int64."""
    perm = np.empty(n, dtype=np.int64)
    _lib.check(_lib.load().irec_tf_shuffle_perm(int(seed), int(n), perm.ctypes.data_as(ctypes.c_void_p)),
               "irec_tf_shuffle_perm")
    return perm


def build_lut():
    lut = np.empty(_lib.BIG_PRIME, dtype=np.float32)
    _lib.check(_lib.load().irec_build_lut(lut.ctypes.data_as(ctypes.c_void_p)), "irec_build_lut")
    return lut


def philox_uniform_int(seed, n):
    out = np.empty(n, dtype=np.int32)
    _lib.check(_lib.load().irec_philox_uniform_int(int(seed), int(n), out.ctypes.data_as(ctypes.c_void_p)),
               "irec_philox_uniform_int")
    return out


class BlockLayout:
    """Descriptors of the blocks of `n_tensors` latent tensors of `n` dims each, cut into <= block_size slices of the
    shuffled order (coder.py:69-83).  Blocks are listed largest first so the persistent kernels end on short ones;
    `natural` maps (tensor, block) -> row of the descriptor arrays."""

    def __init__(self, device, n_tensors, n, block_size, seed):
        self.n_tensors, self.n, self.block_size, self.seed = n_tensors, n, block_size, seed
        bs = n if block_size is None else int(block_size)
        if bs < 1:
            raise ValueError("block_size must be >= 1")
        starts = np.arange(0, n, bs, dtype=np.int64)
        dims = np.minimum(bs, n - starts).astype(np.int64)
        self.blocks_per_tensor = len(starts)
        base = np.repeat(np.arange(n_tensors, dtype=np.int64) * n, len(starts))
        pos = np.tile(starts, n_tensors)
        dim = np.tile(dims, n_tensors)
        order = np.argsort(-dim, kind="stable")
        self.order = order                              # row r describes natural block order[r]
        self.natural = np.empty_like(order)
        self.natural[order] = np.arange(len(order))     # natural block b sits in row natural[b]
        self.n_blocks = len(order)
        self.max_dim = int(dims.max())
        self.distinct_dims = sorted(set(int(d) for d in dims), reverse=True)
        self.block_base = torch.from_numpy(base[order]).to(device)
        self.block_pos = torch.from_numpy(pos[order].astype(np.int32)).to(device)
        self.block_dim = torch.from_numpy(dim[order].astype(np.int32)).to(device)
        if block_size is None:
            self.perm_host = None
            self.perm = None
        else:
            self.perm_host = tf_shuffle_perm(seed, n)
            self.perm = torch.from_numpy(self.perm_host.astype(np.int32)).to(device)


    def natural_dev(self):
        """int32 device copy of `natural` (row of block j of tensor i at [i * blocks_per_tensor + j])."""
        if getattr(self, "_natural_dev", None) is None:
            self._natural_dev = torch.from_numpy(self.natural.astype(np.int32)).to(self.block_base.device)
        return self._natural_dev

    def packed_index(self, n_calls):
        """int64 device index into the rows of `n_calls` read-backs of this layout stacked call after call: position
        (image i, call r, block j) -> r * n_blocks + natural[i * blocks_per_tensor + j] (PendingCode.gather_packed)."""
        cache = self.__dict__.setdefault("_packed_index", {})
        if n_calls not in cache:
            nat = self.natural.reshape(self.n_tensors, 1, self.blocks_per_tensor).astype(np.int64)
            g = nat + (np.arange(n_calls, dtype=np.int64) * self.n_blocks).reshape(1, n_calls, 1)
            cache[n_calls] = torch.from_numpy(np.ascontiguousarray(g.reshape(-1))).to(self.block_base.device)
        return cache[n_calls]

    def subset(self, rows):
        """The layout restricted to `rows` (indices into this layout's row order): what ONE rank codes when the blocks of a
        call are spread over several GPUs (irec/sharding.py).  Same tensors, same permutation."""
        rows = np.asarray(rows, dtype=np.int64)
        sub = object.__new__(BlockLayout)
        sub.n_tensors, sub.n, sub.block_size, sub.seed = self.n_tensors, self.n, self.block_size, self.seed
        sub.blocks_per_tensor = self.blocks_per_tensor
        sub.order = self.order[rows]
        sub.natural = None                       # a subset has no (tensor, block) -> row map: the parent reassembles
        sub._natural_dev = None
        sub.n_blocks = len(rows)
        idx = torch.as_tensor(rows, device=self.block_base.device)
        sub.block_base, sub.block_pos, sub.block_dim = self.block_base[idx], self.block_pos[idx], self.block_dim[idx]
        sub.max_dim = self.max_dim               # (the scratch and the plan are sized for the parent's largest block)
        sub.distinct_dims = self.distinct_dims
        sub.perm_host, sub.perm = self.perm_host, self.perm
        return sub

    def element_index(self, rows, width):
        """[len(rows), width] int64: flat position (over all tensors) of element i of block `rows[r]` in the shuffled order
        (Coder.split, coder.py:62-83), -1 past the block's end."""
        rows = np.asarray(rows, dtype=np.int64)
        base = self.block_base.cpu().numpy()[rows]
        pos = self.block_pos.cpu().numpy()[rows].astype(np.int64)
        dim = self.block_dim.cpu().numpy()[rows].astype(np.int64)
        i = np.arange(width, dtype=np.int64)[None, :]
        inside = i < dim[:, None]
        at = np.where(inside, pos[:, None] + i, 0)
        src = self.perm_host[at] if self.perm_host is not None else at
        return np.where(inside, base[:, None] + src, -1)


class Engine:
    """`lut` (optional, float32 [10007]): the caller's own table of Normal(0,1).quantile(float32(k)/10007) -- the values
    `dist.quantile` takes at beam_search_coder.py:48-49 -- instead of the library's restatement of TFP's float32 ndtri
    (irec_create_ex, include/irec.h).  An engine built this way is private to its creator: pass it to a coder as
    `BeamSearchCoder(..., engine=...)`; `get_engine()` keeps handing out the default-table engine of the device.
    `aux_ratios` (optional, float32 [n]): fitted auxiliary-variance ratios in place of the power law (irec_create_with)."""

    def __init__(self, device, lut=None, aux_ratios=None):
        if not torch.cuda.is_available():
            raise _lib.IrecLibraryError("irec needs a HIP device (MI355X / gfx950); there is no CPU fallback")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.IrecLibraryError(f"irec engine needs a cuda (HIP) device, got {device}")
        self.index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", self.index)
        self.lib = _lib.load()
        ctx = ctypes.c_void_p()
        self.lut = None if lut is None else np.ascontiguousarray(np.asarray(lut), dtype=np.float32)   # (kept: a coder with fitted ratios builds a twin context)
        if lut is None and aux_ratios is None:
            _lib.check(self.lib.irec_create(self.index, ctypes.byref(ctx)), "irec_create")
        else:
            tables = _lib.IrecTables(None, None, 0)
            if lut is not None:
                table = np.ascontiguousarray(np.asarray(lut), dtype=np.float32)
                if table.shape != (_lib.BIG_PRIME,):
                    raise ValueError(f"lut must hold {_lib.BIG_PRIME} float32 values (entry k = quantile(k / 10007)), got shape {table.shape}")
                tables.lut10007 = table.ctypes.data
            if aux_ratios is not None:   # fitted auxiliary-variance ratios (extrapolate_auxiliary_ratios=False, coder.py:203-231)
                ratios = np.ascontiguousarray(np.asarray(aux_ratios), dtype=np.float32).reshape(-1)
                tables.aux_ratios, tables.n_aux_ratios = ratios.ctypes.data, int(ratios.size)
            _lib.check(self.lib.irec_create_with(self.index, ctypes.byref(tables), ctypes.byref(ctx)), "irec_create_with")
        self.ctx = ctx
        self.max_partitions = int(self.lib.irec_max_partitions(ctx))
        self._layouts = {}
        self._ws = {}      # one scratch buffer per HIP stream: calls on different streams never share counters / slabs
        self._tls = threading.local()   # table_session(): key of the calling THREAD's previous call of a back-to-back run of twins
        self._dec_ws = {}  # decode scratch (proposal tables of a call) per HIP stream

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.irec_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # ---- caches ----------------------------------------------------------------------------------------------
    def layout(self, n_tensors, n, block_size, seed):
        key = (n_tensors, n, block_size, seed if block_size is not None else None)
        lay = self._layouts.get(key)
        if lay is None:
            if len(self._layouts) > 64:
                self._layouts.clear()
            lay = BlockLayout(self.device, n_tensors, n, block_size, seed)
            self._layouts[key] = lay
        return lay

    def workspace(self, params, max_dim, max_K, n_blocks=None):
        """Scratch of the CURRENT torch stream (the C ABI is re-entrant; the scratch is what two concurrent calls must
        not share: block counter, proposal tables, beam slabs).  Calls of blocks beyond 1024 dims are sized for their own
        block count (irec_encode_workspace_bytes_for: one slab per team the call launches, not per team slot of the device --
        16 GB at 301 056 dims); the buffer of a stream only grows."""
        if n_blocks is not None and max_dim > 1024:
            need = self.lib.irec_encode_workspace_bytes_for(self.ctx, ctypes.byref(params), int(n_blocks), int(max_dim), int(max_K))
        else:
            need = self.lib.irec_encode_workspace_bytes(self.ctx, ctypes.byref(params), int(max_dim), int(max_K))
        if need == 0:
            raise _lib.IrecLibraryError("irec_encode_workspace_bytes rejected the parameters: " +
                                        self.lib.irec_last_error().decode())
        key = int(torch.cuda.current_stream(self.device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            if ws is None and len(self._ws) >= 16:
                self._ws.clear()       # streams come and go: drop the lot rather than grow without bound
            # head zeroed once: IREC_FLAG_REUSE_TABLES trusts the table stamps there, and only this engine's
            # irec_beam_encode calls ever write to the buffer
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            ws[:512].zero_()
            self._ws[key] = ws
        return ws, need

    def table_session(self):
        """Context manager around a run of encode calls issued back to back on ONE stream by ONE thread (the residual blocks
        of a model pass): a call whose parameters, seed, layout sizes and max_K equal the previous call's finds that call's
        proposal tables still in place, so the table kernels are not even launched (IREC_FLAG_TABLES_PRESENT).  Safe
        inside a HIP-graph capture as well: the first call of the captured run keeps its (device-checked) table kernels."""
        eng = self

        class _Session:
            def __enter__(self_):
                self_.outer = getattr(eng._tls, "session", None)
                eng._tls.session = {"key": None}
                return self_

            def __exit__(self_, *exc):
                eng._tls.session = self_.outer
                return False
        return _Session()

    def plan(self, params, lay, max_K, margins=False):
        """What irec_beam_encode launches for this call (kernel names, grid, LDS, table window): irec_encode_plan."""
        params = self.with_table_dims(params, lay)
        if margins:
            params = self.params(params.kl_per_partition, params.n_samples, params.n_beams, params.flags | _lib.IREC_FLAG_MARGINS,
                                 list(params.table_dims), params.table_steps)
        info = _lib.IrecPlanInfo()
        _lib.check(self.lib.irec_encode_plan(self.ctx, ctypes.byref(params), lay.n_blocks, lay.max_dim, int(max_K),
                                             ctypes.byref(info)), "irec_encode_plan")
        return info.as_dict()

    @staticmethod
    def params(kl_per_partition, n_samples, n_beams, flags=0, table_dims=(), table_steps=0):
        dims = list(table_dims)[:4] if len(table_dims) <= 4 else []
        dims = dims + [0] * (4 - len(dims))
        return _lib.IrecParams(float(np.float32(kl_per_partition)), int(n_samples), int(n_beams), int(flags),
                               (ctypes.c_int32 * 4)(*dims), int(table_steps))

    @staticmethod
    def with_table_dims(params, lay):
        """Copy of `params` carrying the layout's distinct block dims (enables the per-call proposal tables)."""
        return Engine.params(params.kl_per_partition, params.n_samples, params.n_beams, params.flags,
                             lay.distinct_dims, params.table_steps)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- device entry points -----------------------------------------------------------------------------------
    def block_kl(self, params, lay, q_loc, q_scale, p_loc, p_scale):
        out_kl = torch.empty(lay.n_blocks, dtype=torch.float32, device=self.device)
        out_K = torch.empty(lay.n_blocks, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.irec_block_kl(self.ctx, ctypes.byref(params), lay.n_blocks, _ptr(lay.block_base),
                                          _ptr(lay.block_pos), _ptr(lay.block_dim), _ptr(lay.perm), _ptr(q_loc),
                                          _ptr(q_scale), _ptr(p_loc), _ptr(p_scale), _ptr(out_kl), _ptr(out_K),
                                          self._stream()), "irec_block_kl")
        return out_kl, out_K

    def encode_blocks_margins(self, params, lay, q_loc, q_scale, p_loc, p_scale, seed, max_K):
        """encode_blocks that also reports how close every block's top-B selections were (irec_beam_encode_ex, include/irec.h):
        returns (K, indices, sample, margin [n_blocks, 4] float32) -- the same K, indices and sample as encode_blocks, bit for bit."""
        for t in (q_loc, q_scale, p_loc, p_scale):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device
            assert t.numel() == lay.n_tensors * lay.n
        params = self.with_table_dims(params, lay)
        params = self.params(params.kl_per_partition, params.n_samples, params.n_beams, params.flags | _lib.IREC_FLAG_MARGINS,
                             list(params.table_dims), params.table_steps)
        out_K = torch.empty(lay.n_blocks, dtype=torch.int32, device=self.device)
        out_idx = torch.empty((lay.n_blocks, max(max_K, 1)), dtype=torch.int32, device=self.device)
        sample = torch.empty_like(q_loc)
        margin = torch.empty((lay.n_blocks, 4), dtype=torch.float32, device=self.device)
        ws, need = self.workspace(params, lay.max_dim, max_K)
        session = getattr(self._tls, "session", None)
        if session is not None:
            session["key"] = None
        _lib.check(self.lib.irec_beam_encode_ex(self.ctx, ctypes.byref(params), lay.n_blocks, _ptr(lay.block_base),
                                                _ptr(lay.block_pos), _ptr(lay.block_dim), lay.max_dim, _ptr(lay.perm),
                                                _ptr(q_loc), _ptr(q_scale), _ptr(p_loc), _ptr(p_scale), int(seed),
                                                int(max_K), _ptr(out_K), _ptr(out_idx), _ptr(sample), _ptr(margin), _ptr(ws),
                                                ws.numel(), self._stream()), "irec_beam_encode_ex")
        return out_K, out_idx, sample, margin

    def encode_blocks(self, params, lay, q_loc, q_scale, p_loc, p_scale, seed, max_K, out=None, order_by_K=False):
        """Asynchronous.  Returns device tensors (K [n_blocks], indices [n_blocks, max_K], sample [like q_loc]),
        rows in `lay` order.
        order_by_K: hand the blocks to the persistent kernel LONGEST FIRST -- K = ceil(KL / Omega) of every block from
        irec_block_kl (beam_search_coder.py:57-59; an HBM-bound pre-pass, ~0.5 % of a batch call), rows sorted by K x dims on
        the device, outputs scattered back to `lay` order -- instead of largest-dim first only: posteriors whose KL differs by
        orders of magnitude between tensors otherwise leave the call waiting for a few long blocks that were handed out late."""
        for t in (q_loc, q_scale, p_loc, p_scale):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device
            assert t.numel() == lay.n_tensors * lay.n
        if order_by_K and lay.n_blocks > 1:
            _, K0 = self.block_kl(params, lay, q_loc, q_scale, p_loc, p_scale)
            work = K0.to(torch.int64).clamp_(min=0) * lay.block_dim.to(torch.int64)
            order = torch.argsort(work, descending=True, stable=True)
            sub = object.__new__(BlockLayout)
            sub.__dict__.update(lay.__dict__)
            sub.block_base, sub.block_pos, sub.block_dim = lay.block_base[order], lay.block_pos[order], lay.block_dim[order]
            sub.natural, sub._natural_dev = None, None
            K2, idx2, sample = self.encode_blocks(params, sub, q_loc, q_scale, p_loc, p_scale, seed, max_K,
                                                  out=None if out is None else (torch.empty_like(out[0]), torch.empty_like(out[1]), out[2]))
            if out is None:
                out = (torch.empty_like(K2), torch.empty_like(idx2), sample)
            out[0][order] = K2
            out[1][order] = idx2
            return out[0], out[1], sample
        params = self.with_table_dims(params, lay)
        if out is None:
            out_K = torch.empty(lay.n_blocks, dtype=torch.int32, device=self.device)
            out_idx = torch.empty((lay.n_blocks, max(max_K, 1)), dtype=torch.int32, device=self.device)
            sample = torch.empty_like(q_loc)
        else:
            out_K, out_idx, sample = out
        ws, need = self.workspace(params, lay.max_dim, max_K, lay.n_blocks)
        session = getattr(self._tls, "session", None)
        if session is not None:
            # EVERY call inside a session moves the key on: one that does not carry REUSE_TABLES (another coder's settings, a
            # direct call) may lay its slabs over the tables and stamps the slots zero, so the next call must not be told the
            # tables are present because an EARLIER twin left them.  The session belongs to the calling thread.
            key = None
            if params.flags & _lib.IREC_FLAG_REUSE_TABLES:
                key = (ws.data_ptr(), int(torch.cuda.current_stream(self.device).cuda_stream), int(seed), int(max_K), lay.n_blocks,
                       lay.max_dim, bytes(params))
                if session["key"] == key:
                    params = self.params(params.kl_per_partition, params.n_samples, params.n_beams,
                                         params.flags | _lib.IREC_FLAG_TABLES_PRESENT, list(params.table_dims), params.table_steps)
            session["key"] = None      # (until the call below has been issued: an error leaves nothing to trust)
        _lib.check(self.lib.irec_beam_encode(self.ctx, ctypes.byref(params), lay.n_blocks, _ptr(lay.block_base),
                                             _ptr(lay.block_pos), _ptr(lay.block_dim), lay.max_dim, _ptr(lay.perm),
                                             _ptr(q_loc), _ptr(q_scale), _ptr(p_loc), _ptr(p_scale), int(seed),
                                             int(max_K), _ptr(out_K), _ptr(out_idx), _ptr(sample), _ptr(ws),
                                             ws.numel(), self._stream()), "irec_beam_encode")
        if session is not None:
            session["key"] = key
        return out_K, out_idx, sample

    def _decode_ws(self, params, max_K):
        """Decode scratch of the current stream (the proposal tables of one call), or None when the call needs none."""
        need = self.lib.irec_decode_workspace_bytes(self.ctx, ctypes.byref(params), int(max_K))
        if not need:
            return None
        key = int(torch.cuda.current_stream(self.device).cuda_stream)
        ws = self._dec_ws.get(key)
        if ws is None or ws.numel() < need:
            if ws is None and len(self._dec_ws) >= 16:
                self._dec_ws.clear()
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._dec_ws[key] = ws
        return ws

    def decode_blocks(self, params, lay, p_loc, p_scale, seed, K, indices, mode="auto"):
        """Asynchronous.  K / indices rows in `lay` order.  mode:
          "auto"    whole tensors staged in LDS (irec_beam_decode_tensors) when `lay` is a complete layout whose tensors fit,
                    else "tables";
          "tensors" / "tensors_fused"   that entry point, with / without the per-call proposal tables;
          "tables"  block-wise, the call's draw evaluated once into proposal tables where that pays (irec_beam_decode_ws);
          "fused"   block-wise, Philox in the kernel (irec_beam_decode with dim hints);
          "legacy"  irec_beam_decode without hints: the round-2 kernel.
        Same outputs, bit for bit."""
        for t in (p_loc, p_scale):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device
        assert K.dtype == torch.int32 and indices.dtype == torch.int32 and indices.is_contiguous()
        sample = torch.empty_like(p_loc)
        max_K = int(indices.shape[1])
        bs_eff = lay.n if lay.block_size is None else int(lay.block_size)
        fits = lay.natural is not None and bool(self.lib.irec_decode_tensors_supported(ctypes.byref(params), lay.n, bs_eff))
        if mode == "auto":
            mode = "tensors" if fits else "tables"
        if mode in ("tensors", "tensors_fused"):
            if not fits:
                raise ValueError("decode mode 'tensors' needs a complete layout whose tensors fit the LDS")
            if mode == "tensors_fused":
                params = self.params(params.kl_per_partition, params.n_samples, params.n_beams,
                                     params.flags | _lib.IREC_FLAG_FUSED_PHILOX, list(params.table_dims), params.table_steps)
            tparams = self.with_table_dims(params, lay)
            ws = self._decode_ws(tparams, max_K)
            _lib.check(self.lib.irec_beam_decode_tensors(self.ctx, ctypes.byref(tparams), lay.n_tensors, lay.n, bs_eff,
                                                         _ptr(lay.natural_dev()), _ptr(lay.perm), _ptr(p_loc), _ptr(p_scale),
                                                         int(seed), max_K, _ptr(K), _ptr(indices), _ptr(sample), _ptr(ws),
                                                         ws.numel() if ws is not None else 0, self._stream()),
                       "irec_beam_decode_tensors")
            return sample
        if mode != "tables":
            if mode == "fused":
                params = self.with_table_dims(params, lay)
            elif mode != "legacy":
                raise ValueError(f"unknown decode mode {mode!r}")
            _lib.check(self.lib.irec_beam_decode(self.ctx, ctypes.byref(params), lay.n_blocks, _ptr(lay.block_base),
                                                 _ptr(lay.block_pos), _ptr(lay.block_dim), _ptr(lay.perm), _ptr(p_loc),
                                                 _ptr(p_scale), int(seed), max_K, _ptr(K), _ptr(indices),
                                                 _ptr(sample), self._stream()), "irec_beam_decode")
            return sample
        params = self.with_table_dims(params, lay)
        ws = self._decode_ws(params, max_K)
        _lib.check(self.lib.irec_beam_decode_ws(self.ctx, ctypes.byref(params), lay.n_blocks, _ptr(lay.block_base),
                                                _ptr(lay.block_pos), _ptr(lay.block_dim), lay.max_dim, _ptr(lay.perm),
                                                _ptr(p_loc), _ptr(p_scale), int(seed), max_K, _ptr(K), _ptr(indices),
                                                _ptr(sample), _ptr(ws), ws.numel() if ws is not None else 0,
                                                self._stream()), "irec_beam_decode_ws")
        return sample

    # ---- test hooks ------------------------------------------------------------------------------------------------
    def device_uniform_int(self, seed, n):
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.irec_device_uniform_int(self.ctx, int(seed), int(n), _ptr(out), self._stream()),
                   "irec_device_uniform_int")
        return out

    def test_proposal_table(self, seed, n_samples, dim, n_steps):
        dp = (dim + 3) // 4 * 4
        out = torch.zeros((n_steps, n_samples, dp), dtype=torch.int16, device=self.device)
        _lib.check(self.lib.irec_test_proposal_table(self.ctx, int(seed), int(n_samples), int(dim), int(n_steps), _ptr(out),
                                                     self._stream()), "irec_test_proposal_table")
        return out

    def test_select(self, scores, n_select, n_beams_cur, key_offset=0, quick=False):
        n = scores.numel()
        keys = torch.empty(n + key_offset, dtype=torch.int32, device=self.device)[key_offset:]   # (offset: a 4-byte-aligned key array)
        sel = torch.empty((n_select, 2), dtype=torch.int32, device=self.device)
        fn = self.lib.irec_test_select_quick if quick else self.lib.irec_test_select   # (the round-4 form of the selection)
        _lib.check(fn(self.ctx, _ptr(scores), n, int(n_select), int(n_beams_cur), _ptr(keys), _ptr(sel), self._stream()),
                   "irec_test_select")
        return sel

    def test_reduce_scatter(self, x, scoring_form=False):
        width = x.shape[1] + (1 if scoring_form else 0)    # 21 = the scoring loop's form of the 20-value reduce-scatter
        out = torch.zeros(128, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.irec_test_reduce_scatter(self.ctx, _ptr(x), _ptr(out), int(width), self._stream()),
                   "irec_test_reduce_scatter")
        return out


def get_engine(device=None):
    """One engine (irec_context) per HIP device of this process."""
    if device is None:
        if not torch.cuda.is_available():
            raise _lib.IrecLibraryError("irec needs a HIP device (MI355X / gfx950); there is no CPU fallback")
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    eng = _engines.get(idx)
    if eng is None:
        eng = Engine(torch.device("cuda", idx))
        _engines[idx] = eng
    return eng
