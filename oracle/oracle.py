"""ctypes front end of the CPU oracle (oracle/irec_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (relative-entropy-coding_amd/) never does.  Parity status: see the header of irec_oracle.c
("parity unpinned" against real TensorFlow; pinned by KATs + self-generated golden fixtures).

Reference call stack restated here (host side of the block loop):
    GaussianCoder.encode   /root/reference/rec/coding/coder.py:412-457
    GaussianCoder.decode   /root/reference/rec/coding/coder.py:459-491
    Coder.split / merge    /root/reference/rec/coding/coder.py:38-122
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# IREC_ORACLE_LIB_PATH: load another build of the same source by path (scripts/sanitize_oracle.sh: the ASan + UBSan build) --
# the library in this directory is never overwritten
_SO = os.environ.get("IREC_ORACLE_LIB_PATH") or os.path.join(_HERE, "libirec_oracle.so")

CANONICAL = 0
LITERAL = 1
P = 10007


def build(force=False):
    src = os.path.join(_HERE, "irec_oracle.c")
    if os.environ.get("IREC_ORACLE_LIB_PATH"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libirec_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        if not L.irec_oracle_cpu_has_fma():
            raise RuntimeError("oracle was built with -mfma but this CPU has no FMA")
        f32p = ctypes.POINTER(ctypes.c_float)
        i32p = ctypes.POINTER(ctypes.c_int32)
        i64p = ctypes.POINTER(ctypes.c_int64)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        L.irec_oracle_philox4x32.argtypes = [u32p, u32p, u32p]
        L.irec_oracle_uniform_int.argtypes = [ctypes.c_int64, ctypes.c_int64, i32p]
        L.irec_oracle_det_log.argtypes = [ctypes.c_double]
        L.irec_oracle_det_log.restype = ctypes.c_double
        L.irec_oracle_ndtri_f32.argtypes = [ctypes.c_float]
        L.irec_oracle_ndtri_f32.restype = ctypes.c_float
        L.irec_oracle_build_lut.argtypes = [f32p]
        L.irec_oracle_set_lut.argtypes = [f32p]
        L.irec_oracle_py_first_randint31.argtypes = [ctypes.c_int64]
        L.irec_oracle_py_first_randint31.restype = ctypes.c_int64
        L.irec_oracle_py_randint31_nth.argtypes = [ctypes.c_int64, ctypes.c_int64]
        L.irec_oracle_py_randint31_nth.restype = ctypes.c_int64
        u64p = ctypes.POINTER(ctypes.c_uint64)
        L.irec_oracle_tf_get_seed.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                              u64p, u64p]
        L.irec_oracle_tf_get_seed.restype = ctypes.c_int
        L.irec_oracle_tf_uniform_float_pair.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, f32p]
        L.irec_oracle_tf_uniform_int_pair.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int32,
                                                      ctypes.c_int32, ctypes.c_int64, i32p]
        L.irec_oracle_tf_normal_pair.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, f32p]
        L.irec_oracle_tf_shuffle_perm.argtypes = [ctypes.c_int64, ctypes.c_int64, i64p]
        L.irec_oracle_simple_hash.argtypes = [i32p, ctypes.c_int]
        L.irec_oracle_simple_hash.restype = ctypes.c_int32
        L.irec_oracle_set_aux_ratios.argtypes = [f32p, ctypes.c_int]
        L.irec_oracle_aux_ratio.argtypes = [ctypes.c_int]
        L.irec_oracle_aux_ratio.restype = ctypes.c_float
        L.irec_oracle_block_kl.argtypes = [ctypes.c_int, ctypes.c_int, f32p, f32p, f32p, f32p]
        L.irec_oracle_block_kl.restype = ctypes.c_float
        L.irec_oracle_num_aux.argtypes = [ctypes.c_float, ctypes.c_float]
        L.irec_oracle_num_aux.restype = ctypes.c_int32
        L.irec_oracle_encode_block.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               f32p, f32p, f32p, f32p, ctypes.c_int64, ctypes.c_int32, i32p, f32p,
                                               i32p, f32p]
        L.irec_oracle_encode_block.restype = ctypes.c_int32
        L.irec_oracle_encode_block_ex.argtypes = L.irec_oracle_encode_block.argtypes + [f32p]
        L.irec_oracle_encode_block_ex.restype = ctypes.c_int32
        L.irec_oracle_encode_blocks_omp_ex.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                       i32p, i64p, f32p, f32p, f32p, f32p, ctypes.c_int64, ctypes.c_int32,
                                                       i32p, i32p, f32p, f32p, ctypes.c_int]
        L.irec_oracle_encode_blocks_omp_ex.restype = ctypes.c_int
        L.irec_oracle_encode_blocks_omp.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                    i32p, i64p, f32p, f32p, f32p, f32p, ctypes.c_int64, ctypes.c_int32,
                                                    i32p, i32p, f32p, ctypes.c_int]
        L.irec_oracle_encode_blocks_omp.restype = ctypes.c_int
        L.irec_oracle_decode_block.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p, f32p, i32p,
                                               ctypes.c_int32, ctypes.c_int64, f32p]
        L.irec_oracle_tf_random_normal.argtypes = [ctypes.c_int64, ctypes.c_int64, f32p]
        L.irec_oracle_importance_n_samples.argtypes = [ctypes.c_double]
        L.irec_oracle_importance_n_samples.restype = ctypes.c_int64
        L.irec_oracle_tf_stateless_normal.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, f32p]
        L.irec_oracle_importance_encode.argtypes = [f32p, f32p, f32p, f32p, ctypes.c_int64, ctypes.c_double, ctypes.c_double,
                                                    ctypes.c_int64, f32p]
        L.irec_oracle_importance_encode.restype = ctypes.c_int64
        L.irec_oracle_importance_decode.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, f32p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def n_samples(kl_per_partition, extra_samples=1.0):
    """beam_search_coder.py:29"""
    return int(np.exp(kl_per_partition * extra_samples))


def philox4x32(key, ctr):
    k = np.asarray(key, dtype=np.uint32)
    c = np.asarray(ctr, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().irec_oracle_philox4x32(_p(k, ctypes.c_uint32), _p(c, ctypes.c_uint32), _p(o, ctypes.c_uint32))
    return o


def uniform_int(seed, n):
    o = np.zeros(n, dtype=np.int32)
    lib().irec_oracle_uniform_int(int(seed), n, _p(o, ctypes.c_int32))
    return o


def det_log(x):
    return lib().irec_oracle_det_log(float(x))


def build_lut():
    o = np.zeros(P, dtype=np.float32)
    lib().irec_oracle_build_lut(_p(o, ctypes.c_float))
    return o


def set_lut(lut=None):
    """Injects a quantile table into the coder functions (the oracle-side twin of the product's irec_create_ex): float32
    [10007], lut[k] = Normal(0,1).quantile(float32(k)/10007), entry 0 unused.  None restores the restated table."""
    if lut is None:
        lib().irec_oracle_set_lut(None)
        return
    a = np.ascontiguousarray(lut, dtype=np.float32)
    assert a.shape == (P,), a.shape
    lib().irec_oracle_set_lut(_p(a, ctypes.c_float))


def py_first_randint31(seed):
    return lib().irec_oracle_py_first_randint31(int(seed))


class TfEagerRandom:
    """The seed plumbing of TF 2.x eager stateful random ops as the oracle restates it, written as
    the little state machine it is so that the eager outputs printed in TensorFlow's public API docs can be replayed:
    tf.random.set_seed(g) re-creates random.Random(g) and clears the kernel cache; an op WITHOUT an op seed takes the next
    randint of that generator (a fresh kernel every time); an op WITH one re-uses the cached kernel of its (seed, seed2,
    shape, dtype) attributes, whose Philox counter each call advances by 256 blocks per output element."""

    def __init__(self, global_seed=None):
        self.set_seed(global_seed)

    def set_seed(self, global_seed):
        self.g = global_seed
        self.n_auto = 0
        self.skips = {}

    def _stream(self, kind, n, op_seed):
        s1, s2 = ctypes.c_uint64(), ctypes.c_uint64()
        rc = lib().irec_oracle_tf_get_seed(self.g is not None, int(self.g or 0), op_seed is not None, int(op_seed or 0),
                                           self.n_auto, ctypes.byref(s1), ctypes.byref(s2))
        if rc != 0:
            raise ValueError("(None, None): TensorFlow seeds this op non-deterministically")
        if op_seed is None and self.g is not None:
            self.n_auto += 1
        key = (kind, s1.value, s2.value, n)
        skip = self.skips.get(key, 0)
        self.skips[key] = skip + 256 * n
        return s1.value, s2.value, skip

    def uniform(self, n, seed=None):
        """tf.random.uniform([n], seed=seed) (float32, [0, 1))"""
        s1, s2, skip = self._stream("uf", n, seed)
        o = np.zeros(n, dtype=np.float32)
        lib().irec_oracle_tf_uniform_float_pair(s1, s2, skip, n, _p(o, ctypes.c_float))
        return o

    def uniform_int(self, n, minval, maxval, seed=None):
        """tf.random.uniform([n], minval, maxval, dtype=tf.int32, seed=seed)"""
        s1, s2, skip = self._stream(("ui", minval, maxval), n, seed)
        o = np.zeros(n, dtype=np.int32)
        lib().irec_oracle_tf_uniform_int_pair(s1, s2, skip, int(minval), int(maxval), n, _p(o, ctypes.c_int32))
        return o

    def normal(self, n, seed=None):
        """tf.random.normal([n], seed=seed)"""
        s1, s2, skip = self._stream("n", n, seed)
        o = np.zeros(n, dtype=np.float32)
        lib().irec_oracle_tf_normal_pair(s1, s2, skip, n, _p(o, ctypes.c_float))
        return o


def tf_uniform_float(global_seed, op_seed, n):
    """tf.random.set_seed(global_seed); tf.random.uniform([n], seed=op_seed) -- either seed may be None."""
    return TfEagerRandom(global_seed).uniform(n, seed=op_seed)


def tf_shuffle_perm(seed, n):
    o = np.zeros(n, dtype=np.int64)
    lib().irec_oracle_tf_shuffle_perm(int(seed), n, _p(o, ctypes.c_int64))
    return o


def simple_hash(idx):
    a = np.ascontiguousarray(idx, dtype=np.int32)
    return lib().irec_oracle_simple_hash(_p(a, ctypes.c_int32), len(a))


_aux_keep = None


def set_aux_ratios(ratios=None):
    """Fitted auxiliary-variance ratios (extrapolate_auxiliary_ratios=False, coder.py:203-231) for every later call; None: the power law."""
    global _aux_keep
    if ratios is None:
        _aux_keep = None
        lib().irec_oracle_set_aux_ratios(None, 0)
    else:
        _aux_keep = np.ascontiguousarray(ratios, dtype=np.float32)      # (the C side keeps the pointer)
        lib().irec_oracle_set_aux_ratios(_p(_aux_keep, ctypes.c_float), int(_aux_keep.size))


def aux_ratio(i):
    return lib().irec_oracle_aux_ratio(int(i))


def block_kl(mq, sq, mp, sp, mode=CANONICAL):
    mq, sq, mp, sp = map(_f32, (mq, sq, mp, sp))
    return lib().irec_oracle_block_kl(mode, mq.size, _p(mq, ctypes.c_float), _p(sq, ctypes.c_float),
                                      _p(mp, ctypes.c_float), _p(sp, ctypes.c_float))


def num_aux(kl, omega):
    return lib().irec_oracle_num_aux(float(kl), float(np.float32(omega)))


def margins_from_trace(trace, S, B):
    """The four floats of irec_beam_encode_ex's out_margin (include/irec.h) from the scores encode_block(trace=True) recorded --
    an independent (numpy) statement of what irec_oracle_encode_block_ex computes in C."""
    K = trace["K"]
    out = np.array([np.inf, 0.0, np.inf, 0.0], dtype=np.float32)
    Bcur = 1
    for t in range(K):
        N = S * Bcur
        Bnew = min(B, N)
        sc = trace["score"][t][:N].astype(np.float32) + np.float32(0.0)        # (-0 counts as +0)
        srt = sc[np.argsort(-sc, kind="stable")]                                  # value descending, ties to the lower flat index
        if t < K - 1:
            if N > Bnew:
                g = np.float32(srt[Bnew - 1] - srt[Bnew])
                if g < out[0]:
                    out[0], out[1] = g, np.abs(srt[Bnew - 1])
        else:
            if N >= 2:
                out[2] = np.float32(srt[0] - srt[1])
            out[3] = np.abs(srt[0])
        Bcur = Bnew
    return out


def encode_block(mq, sq, mp, sp, seed, omega, S, B, mode=CANONICAL, max_K=4096, trace=False, margins=False):
    """BeamSearchCoder.encode_block on one already-permuted block.  Returns (indices list, sample[, trace][, margins [4]])."""
    mq, sq, mp, sp = map(lambda a: _f32(a).reshape(-1), (mq, sq, mp, sp))
    D = mq.size
    idx = np.zeros(max_K, dtype=np.int32)
    sample = np.zeros(D, dtype=np.float32)
    K0 = num_aux(block_kl(mq, sq, mp, sp, mode), omega)
    sel = np.full((max(K0, 1), B, 2), -1, dtype=np.int32) if trace else None
    sc = np.zeros((max(K0, 1), S * B), dtype=np.float32) if trace else None
    mg = np.zeros(4, dtype=np.float32) if margins else None
    K = lib().irec_oracle_encode_block_ex(mode, float(np.float32(omega)), S, B, D, _p(mq, ctypes.c_float),
                                          _p(sq, ctypes.c_float), _p(mp, ctypes.c_float), _p(sp, ctypes.c_float),
                                          int(seed), max_K, _p(idx, ctypes.c_int32), _p(sample, ctypes.c_float),
                                          _p(sel, ctypes.c_int32) if trace else None,
                                          _p(sc, ctypes.c_float) if trace else None,
                                          _p(mg, ctypes.c_float) if margins else None)
    if K > max_K:
        raise ValueError(f"K={K} exceeds max_K={max_K}")
    out = ([int(v) for v in idx[:K]], sample)
    if trace:
        out = out + ({"sel": sel[:K], "score": sc[:K], "K": K},)
    if margins:
        out = out + (mg,)
    return out


def decode_block(mp, sp, indices, seed, S, mode=CANONICAL):
    mp, sp = map(lambda a: _f32(a).reshape(-1), (mp, sp))
    D = mp.size
    idx = np.ascontiguousarray(indices, dtype=np.int32)
    out = np.zeros(D, dtype=np.float32)
    lib().irec_oracle_decode_block(mode, S, D, _p(mp, ctypes.c_float), _p(sp, ctypes.c_float),
                                   _p(idx, ctypes.c_int32), len(idx), int(seed), _p(out, ctypes.c_float))
    return out


def split_blocks(n, block_size):
    """coder.py:69-83: [start, stop) of each block in the permuted order; the last block may be short."""
    return [(i, min(i + block_size, n)) for i in range(0, n, block_size)]


def encode_tensor(q_loc, q_scale, p_loc, p_scale, seed, omega, S, B, block_size=None, mode=CANONICAL):
    """GaussianCoder.encode (coder.py:412-457) for ONE latent tensor (leading batch dim of 1 allowed).
    Returns (indices, sample) -- indices is list[list[int]] per block when block_size is set, else list[int]."""
    shape = np.shape(q_loc)
    mq, sq, mp, sp = map(lambda a: _f32(a).reshape(-1), (q_loc, q_scale, p_loc, p_scale))
    if block_size is None:
        idx, samp = encode_block(mq, sq, mp, sp, seed, omega, S, B, mode)
        return idx, samp.reshape(shape)
    n = mq.size
    perm = tf_shuffle_perm(seed, n)
    out = np.zeros(n, dtype=np.float32)
    indices = []
    for lo, hi in split_blocks(n, block_size):
        g = perm[lo:hi]
        idx, samp = encode_block(mq[g], sq[g], mp[g], sp[g], seed, omega, S, B, mode)
        indices.append(idx)
        out[g] = samp  # merge: inverse permutation (coder.py:111-117)
    return indices, out.reshape(shape)


def encode_tensors_omp(q_loc, q_scale, p_loc, p_scale, seed, omega, S, B, block_size, mode=CANONICAL, max_K=64,
                       n_threads=0, margins=False):
    """CPU-opt baseline: GaussianCoder.encode over a batch [N, n] of latent tensors, OpenMP over the
    N * blocks independent blocks.  Returns (indices[N][blocks][K], sample [N, n], threads used[, margins [N, blocks, 4]])."""
    mq, sq, mp, sp = (np.ascontiguousarray(a, dtype=np.float32).reshape(len(a), -1) for a in (q_loc, q_scale, p_loc, p_scale))
    N, n = mq.shape
    perm = tf_shuffle_perm(seed, n)
    blocks = split_blocks(n, block_size)
    dims = np.tile(np.array([hi - lo for lo, hi in blocks], dtype=np.int32), N)
    offs = (np.repeat(np.arange(N, dtype=np.int64) * n, len(blocks)) + np.tile(np.array([lo for lo, _ in blocks], dtype=np.int64), N))
    pm = [np.ascontiguousarray(a[:, perm]).reshape(-1) for a in (mq, sq, mp, sp)]   # split: gather through perm
    out_K = np.zeros(len(dims), dtype=np.int32)
    out_idx = np.zeros((len(dims), max_K), dtype=np.int32)
    out_s = np.zeros(N * n, dtype=np.float32)
    mg = np.zeros((len(dims), 4), dtype=np.float32) if margins else None
    used = lib().irec_oracle_encode_blocks_omp_ex(mode, float(np.float32(omega)), S, B, len(dims), _p(dims, ctypes.c_int32),
                                                  _p(offs, ctypes.c_int64), *(_p(a, ctypes.c_float) for a in pm), int(seed),
                                                  max_K, _p(out_K, ctypes.c_int32), _p(out_idx, ctypes.c_int32),
                                                  _p(out_s, ctypes.c_float), _p(mg, ctypes.c_float) if margins else None,
                                                  int(n_threads))
    if (out_K > max_K).any():
        raise ValueError(f"K={int(out_K.max())} exceeds max_K={max_K}")
    sample = np.zeros((N, n), dtype=np.float32)
    sample[:, perm] = out_s.reshape(N, n)                                             # merge: inverse permutation
    nb = len(blocks)
    indices = [[[int(v) for v in out_idx[i * nb + j, :out_K[i * nb + j]]] for j in range(nb)] for i in range(N)]
    if margins:
        return indices, sample, used, mg.reshape(N, nb, 4)
    return indices, sample, used


def decode_tensor(p_loc, p_scale, indices, seed, S, block_size=None, mode=CANONICAL):
    """GaussianCoder.decode (coder.py:459-491)."""
    shape = np.shape(p_loc)
    mp, sp = map(lambda a: _f32(a).reshape(-1), (p_loc, p_scale))
    if block_size is None:
        return decode_block(mp, sp, indices, seed, S, mode).reshape(shape)
    n = mp.size
    perm = tf_shuffle_perm(seed, n)
    out = np.zeros(n, dtype=np.float32)
    for (lo, hi), idx in zip(split_blocks(n, block_size), indices):
        g = perm[lo:hi]
        out[g] = decode_block(mp[g], sp[g], idx, seed, S, mode)
    return out.reshape(shape)


def codelength(indices, S):
    """beam_search_coder.py:150-151 (nats)."""
    return len(indices) * np.log(S)


def synthetic_latent(image_id, n, rng_base=1234):
    """SURVEY.md §8d synthetic posterior/prior statistics for one latent tensor of n dims."""
    rng = np.random.default_rng(rng_base + image_id)
    mp = rng.normal(0.0, 1.0, n)
    lsp = rng.normal(0.0, 0.25, n)
    sp = np.exp(lsp)
    mq = mp + sp * rng.normal(0.0, 0.2, n)
    sq = np.exp(lsp - np.abs(rng.normal(0.0, 0.05, n)))
    return tuple(a.astype(np.float32) for a in (mq, sq, mp, sp))


# ---- importance sampler (config 1 plumbing; rec/coding/importance_sampling.py) ---------------------------------------
def tf_random_normal(seed, count):
    o = np.zeros(count, dtype=np.float32)
    lib().irec_oracle_tf_random_normal(int(seed), int(count), _p(o, ctypes.c_float))
    return o


def importance_n_samples(coding_bits):
    return int(lib().irec_oracle_importance_n_samples(float(coding_bits)))


def tf_stateless_normal(seed0, seed1, count):
    o = np.zeros(count, dtype=np.float32)
    lib().irec_oracle_tf_stateless_normal(int(seed0), int(seed1), int(count), _p(o, ctypes.c_float))
    return o


def importance_encode(t_loc, t_scale, p_loc, p_scale, coding_bits, seed, alpha=float("inf")):
    tl, ts, pl, ps = (_f32(a).reshape(-1) for a in (t_loc, t_scale, p_loc, p_scale))
    out = np.zeros_like(tl)
    idx = lib().irec_oracle_importance_encode(_p(tl, ctypes.c_float), _p(ts, ctypes.c_float), _p(pl, ctypes.c_float),
                                              _p(ps, ctypes.c_float), tl.size, float(coding_bits), float(alpha), int(seed),
                                              _p(out, ctypes.c_float))
    return int(idx), out.reshape(np.shape(t_loc))


def importance_decode(p_loc, p_scale, index, seed):
    pl, ps = (_f32(a).reshape(-1) for a in (p_loc, p_scale))
    out = np.zeros_like(pl)
    lib().irec_oracle_importance_decode(_p(pl, ctypes.c_float), _p(ps, ctypes.c_float), pl.size, int(index), int(seed),
                                        _p(out, ctypes.c_float))
    return out.reshape(np.shape(p_loc))
