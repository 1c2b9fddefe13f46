"""The `.rec` container of the reference (rec/io/utils.py:7-215): struct header + arithmetic-coded streams of
(number of auxiliary variables per block, sample indices) per residual block.  Same signatures, same bytes."""
import ctypes
import struct

import numpy as np

from .. import _lib
from .entropy_coding import ArithmeticCoder

STATIC_HEADER = "IIIIIHHHH"  # seed, block_size, max_index, h, w, c, 2 flags, num_res_blocks (utils.py:81-95)


def _index_counts(max_index):
    counts = np.ones(max_index + 1, dtype=np.int64)   # utils.py:31-35
    counts[1:] += 1000
    return counts


def _nav_counts(nav_max):
    counts = np.ones(nav_max + 2, dtype=np.int64)     # utils.py:41-47
    counts[1:] += 100
    return counts


def _to_message(msg):
    return np.concatenate([np.asarray(msg, dtype=np.int64).reshape(-1) + 1, [0]], axis=0)  # utils.py:58-59


def _pack(code):
    """'1' + code as a big-endian integer in ceil(len/8) bytes (utils.py:66-72,100-106)."""
    lib = _lib.load()
    bits = np.frombuffer("".join(code).encode("ascii"), dtype=np.uint8) if len(code) else np.zeros(0, np.uint8)
    out = np.empty((bits.size + 1 + 7) // 8, dtype=np.uint8)
    n = lib.irec_rec_pack_bits(bits.ctypes.data if bits.size else None, bits.size, out.ctypes.data, out.size)
    if n < 0:
        raise ValueError("irec_rec_pack_bits failed")
    return out[:n].tobytes()


def _unpack(data):
    lib = _lib.load()
    buf = np.frombuffer(data, dtype=np.uint8)
    out = np.empty(max(buf.size * 8, 1), dtype=np.uint8)
    n = lib.irec_rec_unpack_bits(buf.ctypes.data if buf.size else None, buf.size, out.ctypes.data, out.size)
    if n < 0:
        raise ValueError("corrupt .rec stream (no marker bit)")
    return out[:n].tobytes().decode("ascii")


def write_compressed_code(file_path, seed, image_shape, block_size, block_indices, max_index,
                          num_aux_var_counts_file=None, index_counts_file=None):
    """rec/io/utils.py:7-106.  block_indices: per residual block, per coded block, the list of sample indices."""
    if len(image_shape) != 3:
        raise ValueError(f"Image shape must be rank 3, but was {image_shape}!")
    img_h, img_w, img_c = image_shape
    num_res_blocks = len(block_indices)
    num_blocks = list(map(len, block_indices))
    num_aux_vars = [list(map(len, block)) for block in block_indices]
    flattened = [np.concatenate([np.asarray(b, dtype=np.int64).reshape(-1) for b in block], axis=0)
                 for block in block_indices]
    index_counts = _index_counts(max_index) if index_counts_file is None else np.load(index_counts_file)
    for f in flattened:
        if f.size and (f.min() < 0 or f.max() + 1 > len(index_counts) - 1):
            # the reference overflows its count table silently here (max_index=20 with S=36)
            raise ValueError(f"index {int(f.max())} does not fit max_index={len(index_counts) - 1}")
    if num_aux_var_counts_file is None:
        num_aux_var_maxes = [int(np.max(nav)) for nav in num_aux_vars]
        num_aux_var_counts = [_nav_counts(m) for m in num_aux_var_maxes]
    else:
        num_aux_var_counts = np.load(num_aux_var_counts_file, allow_pickle=True)
        num_aux_var_maxes = [-1] * num_res_blocks
    nav_coders = [ArithmeticCoder(c, precision=32) for c in num_aux_var_counts]
    index_coder = ArithmeticCoder(index_counts, precision=32)
    nav_bytes = [_pack(coder.encode(_to_message(nav))) for nav, coder in zip(num_aux_vars, nav_coders)]
    index_bytes = [_pack(index_coder.encode(_to_message(ix))) for ix in flattened]
    header = struct.pack(f"{STATIC_HEADER}{num_res_blocks}I{num_res_blocks}I{num_res_blocks}I{num_res_blocks}I",
                         seed, block_size, max_index, img_h, img_w, img_c,
                         1 - int(num_aux_var_counts_file is None), 1 - int(index_counts_file is None), num_res_blocks,
                         *num_blocks, *[len(b) for b in nav_bytes], *[len(b) for b in index_bytes],
                         *[m & 0xFFFFFFFF for m in num_aux_var_maxes])
    with open(file_path, "wb") as rec_file:
        rec_file.write(header)
        for b in nav_bytes:
            rec_file.write(b)
        for b in index_bytes:
            rec_file.write(b)


def read_compressed_code(file_path, static_header_size=28, num_aux_var_counts_file=None, index_counts_file=None):
    """rec/io/utils.py:109-215.  Returns (seed, image_shape, block_size, block_indices)."""
    with open(file_path, "rb") as rec_file:
        info = struct.unpack(STATIC_HEADER, rec_file.read(static_header_size))
        seed, block_size, max_index = info[0], info[1], info[2]
        image_shape = tuple(info[3:6])
        use_nav_file, use_index_file, num_res_blocks = bool(info[6]), bool(info[7]), info[8]
        if use_index_file and index_counts_file is None:
            raise ValueError("The compressed file is using empirical index counts, but no counts file was supplied!")
        if use_nav_file and num_aux_var_counts_file is None:
            raise ValueError("The compressed file is using empirical num_aux_var counts, but no counts file was supplied!")
        fmt = f"{num_res_blocks}I{num_res_blocks}I{num_res_blocks}I{num_res_blocks}I"
        dyn = struct.unpack(fmt, rec_file.read(struct.calcsize(fmt)))
        nav_lens = dyn[num_res_blocks:2 * num_res_blocks]
        index_lens = dyn[2 * num_res_blocks:3 * num_res_blocks]
        nav_maxes = dyn[3 * num_res_blocks:]
        nav_codes = [_unpack(rec_file.read(n)) for n in nav_lens]
        index_codes = [_unpack(rec_file.read(n)) for n in index_lens]
    index_counts = np.load(index_counts_file) if use_index_file else _index_counts(max_index)
    if use_nav_file:
        nav_counts = np.load(num_aux_var_counts_file, allow_pickle=True)
    else:
        nav_counts = [_nav_counts(m) for m in nav_maxes]
    nav_coders = [ArithmeticCoder(c, precision=32) for c in nav_counts]
    index_coder = ArithmeticCoder(index_counts, precision=32)

    def from_message(msg):
        return np.array(msg, dtype=np.int64)[:-1] - 1

    num_aux_vars = [from_message(c.decode_fast(code)) for c, code in zip(nav_coders, nav_codes)]
    flattened = [from_message(index_coder.decode_fast(code)) for code in index_codes]
    block_indices = []
    for nav, vec in zip(num_aux_vars, flattened):
        bounds = np.cumsum(np.concatenate([[0], nav], axis=0))
        block_indices.append([vec[bounds[i - 1]:bounds[i]].tolist() for i in range(1, len(bounds))])
    return seed, image_shape, block_size, block_indices
