"""ArithmeticCoder with the reference's interface (rec/io/entropy_coding.pyx:19-302), backed by the C++ coder of
libirec_hip.so (csrc/irec_io.cpp).  Codes are lists of '0'/'1' characters, exactly as the reference returns them."""
import ctypes

import numpy as np

from .. import _lib


class ArithmeticCoder(object):
    def __init__(self, P, precision=32):
        """entropy_coding.pyx:21-46."""
        self._P = np.ascontiguousarray(P, dtype=np.int64)
        if self._P.ndim != 1 or self._P.size < 1 or (self._P < 1).any():
            raise ValueError("P must be a 1-D array of counts >= 1")
        self._precision = int(precision)
        self.C = np.concatenate([[0], np.cumsum(self._P)[:-1]]).astype(np.int64)
        self.D = np.cumsum(self._P).astype(np.int64)
        self.R = int(self.D[-1])

    def encode(self, message):
        """entropy_coding.pyx:51-121."""
        lib = _lib.load()
        msg = np.ascontiguousarray(message, dtype=np.int64).reshape(-1)
        cap = 64 + 40 * (msg.size + 1)
        while True:
            out = np.empty(cap, dtype=np.uint8)
            n = ctypes.c_int64(0)
            st = lib.irec_ac_encode(self._P.ctypes.data, self._P.size, msg.ctypes.data, msg.size, self._precision,
                                    out.ctypes.data, cap, ctypes.byref(n))
            if st == 0:
                return list(out[:n.value].tobytes().decode("ascii"))
            if n.value > cap:
                cap = n.value
                continue
            raise ValueError(lib.irec_io_last_error().decode())

    def decode_fast(self, code, verbose=False):
        """entropy_coding.pyx:213-302."""
        lib = _lib.load()
        bits = np.frombuffer("".join(code).encode("ascii"), dtype=np.uint8) if len(code) else np.zeros(0, np.uint8)
        cap = max(64, len(bits) * 4)
        while True:
            out = np.empty(cap, dtype=np.int64)
            n = ctypes.c_int64(0)
            st = lib.irec_ac_decode(self._P.ctypes.data, self._P.size, bits.ctypes.data if bits.size else None,
                                    bits.size, self._precision, out.ctypes.data, cap, ctypes.byref(n))
            if st == 0:
                return [int(v) for v in out[:n.value]]
            if n.value > cap:
                cap = n.value
                continue
            raise ValueError(lib.irec_io_last_error().decode())

    def decode(self, code):
        """entropy_coding.pyx:125-209 (linear symbol search; same result as decode_fast)."""
        return self.decode_fast(code)
