"""The `.rec` container (wire format of the reference, rec/io/utils.py:7-215).

Layout, all little-endian native `struct` fields as the reference writes them:

    static header  'IIIIIHHHH'   seed, block_size, max_index, height, width, channels,
                                 uses_num_aux_var_counts_file, uses_index_counts_file, R   (R = residual blocks)
    dynamic header 4 x R x 'I'   blocks per residual block | byte length of each "count" stream |
                                 byte length of each index stream | largest partition count per residual block
    R count streams, then R index streams

A stream is an arithmetic-coded message (values + 1, terminated by symbol 0; model = counts of 1 for the terminator and
101 / 1001 for everything else) with a marker 1 bit in front, right-aligned in big-endian bytes.  The entropy coder
is the C++ one of libirec_hip.so (irec.io.ArithmeticCoder).  Bytes are identical to the reference's writer
(tests/test_rec_io.py).
"""
import itertools
import struct
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

from .. import _lib
from .entropy_coding import ArithmeticCoder

_STATIC = struct.Struct("IIIIIHHHH")


def _uniform_model(n_values: int, weight: int) -> np.ndarray:
    """Terminator count 1, every value symbol `1 + weight` (utils.py:31-35 with weight 1000, :41-47 with weight 100)."""
    model = np.full(n_values + 1, 1 + weight, dtype=np.int64)
    model[0] = 1
    return model


def _bits_to_bytes(code: Sequence[str]) -> bytes:
    lib = _lib.load()
    bits = np.frombuffer("".join(code).encode("ascii"), dtype=np.uint8) if len(code) else np.zeros(0, np.uint8)
    out = np.empty((bits.size + 8) // 8, dtype=np.uint8)
    n = lib.irec_rec_pack_bits(bits.ctypes.data if bits.size else None, bits.size, out.ctypes.data, out.size)
    if n < 0:
        raise ValueError("irec_rec_pack_bits failed")
    return out[:n].tobytes()


def _bytes_to_bits(data: bytes) -> str:
    lib = _lib.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(max(buf.size * 8, 1), dtype=np.uint8)
    n = lib.irec_rec_unpack_bits(buf.ctypes.data if buf.size else None, buf.size, out.ctypes.data, out.size)
    if n < 0:
        raise ValueError("corrupt .rec stream (no marker bit)")
    return out[:n].tobytes().decode("ascii")


def _encode_stream(model: np.ndarray, values) -> bytes:
    message = np.append(np.asarray(values, dtype=np.int64).reshape(-1) + 1, 0)
    return _bits_to_bytes(ArithmeticCoder(model, precision=32).encode(message))


def _decode_stream(model: np.ndarray, data: bytes) -> np.ndarray:
    message = ArithmeticCoder(model, precision=32).decode_fast(_bytes_to_bits(data))
    return np.asarray(message[:-1], dtype=np.int64) - 1


@dataclass
class RecHeader:
    seed: int
    block_size: int
    max_index: int
    image_shape: Tuple[int, int, int]
    uses_count_file: bool
    uses_index_file: bool
    blocks_per_res_block: List[int]
    count_stream_bytes: List[int]
    index_stream_bytes: List[int]
    max_partitions: List[int]

    def pack(self) -> bytes:
        r = len(self.blocks_per_res_block)
        tail = [*self.blocks_per_res_block, *self.count_stream_bytes, *self.index_stream_bytes,
                *[m & 0xFFFFFFFF for m in self.max_partitions]]
        return _STATIC.pack(self.seed, self.block_size, self.max_index, *self.image_shape, int(self.uses_count_file),
                            int(self.uses_index_file), r) + struct.pack(f"{4 * r}I", *tail)

    @classmethod
    def read(cls, fh, static_header_size=_STATIC.size) -> "RecHeader":
        seed, block_size, max_index, h, w, c, f_counts, f_index, r = _STATIC.unpack(fh.read(static_header_size))
        tail = struct.unpack(f"{4 * r}I", fh.read(16 * r))
        return cls(seed, block_size, max_index, (h, w, c), bool(f_counts), bool(f_index), list(tail[:r]),
                   list(tail[r:2 * r]), list(tail[2 * r:3 * r]), list(tail[3 * r:]))


def _native_encode(seed, image_shape, block_size, block_indices, max_index):
    """The whole container in one C++ call (irec_rec_encode_file): default symbol models only."""
    lib = _lib.load()
    bpr = np.array([len(rb) for rb in block_indices], dtype=np.int32)
    K = np.array([len(ix) for rb in block_indices for ix in rb], dtype=np.int32)
    flat = np.fromiter(itertools.chain.from_iterable(itertools.chain.from_iterable(block_indices)), dtype=np.int32,
                       count=int(K.sum()))
    h, w, c = (int(v) for v in image_shape)
    cap = 64 + 16 * len(bpr) + 4 * (K.size + flat.size) + 64
    while True:
        out = np.empty(cap, dtype=np.uint8)
        n = lib.irec_rec_encode_file(int(seed), int(block_size), int(max_index), h, w, c, len(bpr), bpr.ctypes.data,
                                     K.ctypes.data, flat.ctypes.data if flat.size else None, out.ctypes.data, cap)
        if n < 0:
            raise ValueError(lib.irec_io_last_error().decode())
        if n <= cap:
            return out[:n].tobytes()
        cap = int(n)


def _native_decode(data):
    lib = _lib.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    hdr = np.zeros(9, dtype=np.uint32)
    sizes = np.zeros(3, dtype=np.int64)
    # first try with buffers sized from the header (block counts) and the file size (an index costs >= 2 bits unless
    # max_index is tiny); the library reports the exact sizes if they were short, and only then is the file decoded twice
    n_res = int(np.frombuffer(data[26:28], dtype="<u2")[0]) if len(data) >= 28 else 0
    n_blk = int(np.frombuffer(data[28:28 + 4 * n_res], dtype="<u4").sum()) if len(data) >= 28 + 4 * n_res else 0
    n_blk = min(n_blk, 8 * len(data) + 8)            # (a damaged header: a coded block costs at least a bit of its count stream)
    sizes[:] = (n_res, n_blk, 4 * len(data) + 1024)
    for _attempt in range(2):
        bpr = np.empty(max(int(sizes[0]), 1), dtype=np.int32)
        K = np.empty(max(int(sizes[1]), 1), dtype=np.int32)
        idx = np.empty(max(int(sizes[2]), 1), dtype=np.int32)
        st = lib.irec_rec_decode_file(buf.ctypes.data, buf.size, hdr.ctypes.data, sizes.ctypes.data, bpr.ctypes.data, bpr.size,
                                      K.ctypes.data, K.size, idx.ctypes.data, idx.size)
        if st != _lib.IREC_E_WORKSPACE:
            break
    if st != 0:
        raise ValueError(lib.irec_io_last_error().decode())
    blocks, kb, ib = [], 0, 0
    for r in range(int(sizes[0])):
        rb = []
        for _ in range(int(bpr[r])):
            k = int(K[kb]); kb += 1
            rb.append(idx[ib:ib + k].tolist()); ib += k
        blocks.append(rb)
    return int(hdr[0]), (int(hdr[3]), int(hdr[4]), int(hdr[5])), int(hdr[1]), blocks


def encode_files(seed, image_shape, block_size, K, idx, max_index, n_threads=0):
    """N containers at once from a packed read-back (irec_rec_encode_files): K [N, R, bpt] int32, idx [N, R, bpt, max_K] int32.
    Returns (blob uint8, offsets int64 [N + 1]): file i = blob[offsets[i]:offsets[i + 1]], byte for byte what
    write_compressed_code writes for image i (rec/io/utils.py:7-106, default symbol models)."""
    lib = _lib.load()
    K = np.ascontiguousarray(K, dtype=np.int32)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    n, r, bpt = K.shape
    max_K = idx.shape[3] if idx.ndim == 4 else 0
    assert idx.shape[:3] == K.shape
    h, w, c = (int(v) for v in image_shape)
    offsets = np.zeros(n + 1, dtype=np.int64)
    cap = n * (64 + 16 * r) + 2 * int(K.sum()) + 16 * K.size + 1024
    while True:
        out = np.empty(cap, dtype=np.uint8)
        total = lib.irec_rec_encode_files(int(seed), int(block_size), int(max_index), h, w, c, n, r, bpt, max_K, K.ctypes.data,
                                          idx.ctypes.data if idx.size else None, out.ctypes.data, cap, offsets.ctypes.data,
                                          int(n_threads))
        if total < 0:
            raise ValueError(lib.irec_io_last_error().decode())
        if total <= cap:
            return out[:total], offsets
        cap = int(total)


def decode_files(blob, offsets, n_res_blocks, blocks_per_res, max_K, n_threads=0):
    """The inverse of encode_files (irec_rec_decode_files): (headers [N, 9] uint32 -- seed, block_size, max_index, height,
    width, channels, two flags, R --, K [N, R, bpt], idx [N, R, bpt, max_K] with rows zero-filled past K)."""
    lib = _lib.load()
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = offsets.size - 1
    hdr = np.zeros((n, 9), dtype=np.uint32)
    K = np.zeros((n, n_res_blocks, blocks_per_res), dtype=np.int32)
    idx = np.zeros((n, n_res_blocks, blocks_per_res, max(max_K, 1)), dtype=np.int32)
    st = lib.irec_rec_decode_files(blob.ctypes.data, offsets.ctypes.data, n, int(n_res_blocks), int(blocks_per_res), int(max_K),
                                   hdr.ctypes.data, K.ctypes.data, idx.ctypes.data, int(n_threads))
    if st != 0:
        raise ValueError(lib.irec_io_last_error().decode())
    return hdr, K, idx[..., :max_K]


def write_compressed_code(file_path, seed, image_shape, block_size, block_indices, max_index,
                          num_aux_var_counts_file=None, index_counts_file=None):
    """Same signature as rec/io/utils.py:7.  block_indices[r][k] = sample indices of coded block k of residual block r.
    With the default symbol models (no count files) the container is assembled natively in one call; the per-stream Python
    path below is the reference-shaped one (byte-identical, tests/test_rec_io.py) and serves the count-file variants."""
    if len(image_shape) != 3:
        raise ValueError(f"Image shape must be rank 3, but was {image_shape}!")
    if num_aux_var_counts_file is None and index_counts_file is None and len(block_indices) > 0 and \
            all(len(rb) > 0 for rb in block_indices):
        with open(file_path, "wb") as fh:
            fh.write(_native_encode(seed, image_shape, block_size, block_indices, max_index))
        return
    return _write_compressed_code_py(file_path, seed, image_shape, block_size, block_indices, max_index,
                                     num_aux_var_counts_file, index_counts_file)


def _write_compressed_code_py(file_path, seed, image_shape, block_size, block_indices, max_index,
                              num_aux_var_counts_file=None, index_counts_file=None):
    """The reference-shaped writer: one ArithmeticCoder call per stream (rec/io/utils.py:7-106)."""
    partition_counts = [[len(ix) for ix in res_block] for res_block in block_indices]
    flat_indices = [np.concatenate([np.asarray(ix, dtype=np.int64).reshape(-1) for ix in res_block])
                    for res_block in block_indices]
    index_model = _uniform_model(max_index, 1000) if index_counts_file is None else np.load(index_counts_file)
    for vec in flat_indices:
        if vec.size and (vec.min() < 0 or vec.max() + 1 >= len(index_model)):
            # the reference overruns its count table silently here (max_index=20 with S=36)
            raise ValueError(f"index {int(vec.max())} does not fit max_index={len(index_model) - 1}")
    if num_aux_var_counts_file is None:
        max_partitions = [int(max(pc)) for pc in partition_counts]
        count_models = [_uniform_model(m + 1, 100) for m in max_partitions]
    else:
        count_models = np.load(num_aux_var_counts_file, allow_pickle=True)
        max_partitions = [-1] * len(block_indices)
    count_streams = [_encode_stream(model, pc) for model, pc in zip(count_models, partition_counts)]
    index_streams = [_encode_stream(index_model, vec) for vec in flat_indices]
    header = RecHeader(seed, block_size, max_index, tuple(int(v) for v in image_shape),
                       num_aux_var_counts_file is not None, index_counts_file is not None,
                       [len(rb) for rb in block_indices], [len(s) for s in count_streams],
                       [len(s) for s in index_streams], max_partitions)
    with open(file_path, "wb") as fh:
        fh.write(header.pack())
        fh.writelines(count_streams)
        fh.writelines(index_streams)


def read_compressed_code(file_path, static_header_size=28, num_aux_var_counts_file=None, index_counts_file=None):
    """Same signature and return value as rec/io/utils.py:109: (seed, image_shape, block_size, block_indices)."""
    if static_header_size == 28 and num_aux_var_counts_file is None and index_counts_file is None:
        with open(file_path, "rb") as fh:
            data = fh.read()
        if len(data) >= 28 and data[22:26] == b"\x00\x00\x00\x00":     # neither count-file flag set: default models
            return _native_decode(data)
    return _read_compressed_code_py(file_path, static_header_size, num_aux_var_counts_file, index_counts_file)


def _read_compressed_code_py(file_path, static_header_size=28, num_aux_var_counts_file=None, index_counts_file=None):
    """The reference-shaped reader (rec/io/utils.py:109-216)."""
    with open(file_path, "rb") as fh:
        hdr = RecHeader.read(fh, static_header_size)
        if hdr.uses_index_file and index_counts_file is None:
            raise ValueError("The compressed file is using empirical index counts, but no counts file was supplied!")
        if hdr.uses_count_file and num_aux_var_counts_file is None:
            raise ValueError("The compressed file is using empirical num_aux_var counts, but no counts file was supplied!")
        count_streams = [fh.read(n) for n in hdr.count_stream_bytes]
        index_streams = [fh.read(n) for n in hdr.index_stream_bytes]
    index_model = np.load(index_counts_file) if hdr.uses_index_file else _uniform_model(hdr.max_index, 1000)
    count_models = (np.load(num_aux_var_counts_file, allow_pickle=True) if hdr.uses_count_file
                    else [_uniform_model(m + 1, 100) for m in hdr.max_partitions])
    block_indices = []
    for model, cs, xs in zip(count_models, count_streams, index_streams):
        counts = _decode_stream(model, cs)
        values = _decode_stream(index_model, xs)
        cuts = np.concatenate([[0], np.cumsum(counts)])
        block_indices.append([values[cuts[i]:cuts[i + 1]].tolist() for i in range(len(counts))])
    return hdr.seed, hdr.image_shape, hdr.block_size, block_indices
