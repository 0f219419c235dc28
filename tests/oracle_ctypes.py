"""ctypes binding of oracle/liborc_oracle.so (the CPU oracle -- test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

OK, OUT_OF_SPEC, INVALID_UTF8, DECODE_FLOAT, UNSUPPORTED = 0, 1, 2, 5, 100


class Run(C.Structure):
    _fields_ = [("byte_offset", C.c_uint64), ("byte_length", C.c_uint64), ("kind", C.c_uint32),
                ("bit_width", C.c_uint32), ("length", C.c_uint32), ("reserved", C.c_uint32)]


def _u8(buf):
    a = np.frombuffer(bytes(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    return np.ascontiguousarray(a, dtype=np.uint8)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        u8p, sz, i64p = C.c_void_p, C.c_size_t, C.c_void_p
        lib.orc_oracle_rlev2_i64.argtypes = [u8p, sz, sz, C.c_int, i64p, C.POINTER(sz)]
        lib.orc_oracle_rlev2_runs.argtypes = [u8p, sz, sz, C.POINTER(Run), sz, C.POINTER(sz)]
        lib.orc_oracle_byte_rle.argtypes = [u8p, sz, sz, u8p, C.POINTER(sz)]
        lib.orc_oracle_boolean.argtypes = [u8p, sz, sz, u8p, C.POINTER(C.c_uint64)]
        lib.orc_oracle_float.argtypes = [u8p, sz, sz, sz, C.c_void_p]
        lib.orc_oracle_string_direct.argtypes = [u8p, sz, u8p, sz, sz, C.c_int, i64p, u8p]
        lib.orc_oracle_string_dictionary.argtypes = [i64p, u8p, sz, i64p, sz, i64p, u8p]
        lib.orc_oracle_zigzag.argtypes = [C.c_uint64]
        lib.orc_oracle_zigzag.restype = C.c_int64
        lib.orc_oracle_unpack.argtypes = [u8p, C.c_uint32, sz]
        lib.orc_oracle_unpack.restype = C.c_uint64

    def rlev2(self, stream, n, signed):
        s = _u8(stream)
        out = np.empty(n, dtype=np.int64)
        consumed = C.c_size_t(0)
        rc = self.lib.orc_oracle_rlev2_i64(s.ctypes.data, s.size, n, int(signed), out.ctypes.data,
                                           C.byref(consumed))
        return rc, out, consumed.value

    def rlev1(self, stream, n, signed):
        s = _u8(stream)
        out = np.empty(n, dtype=np.int64)
        consumed = C.c_size_t(0)
        self.lib.orc_oracle_rlev1_i64.argtypes = self.lib.orc_oracle_rlev2_i64.argtypes
        self.lib.orc_oracle_rlev1_i64.restype = C.c_int
        rc = self.lib.orc_oracle_rlev1_i64(s.ctypes.data, s.size, n, int(signed), out.ctypes.data,
                                           C.byref(consumed))
        return rc, out, consumed.value

    def runs(self, stream, n):
        s = _u8(stream)
        cap = max(16, min(n, s.size))  # a run takes at least two stream bytes
        arr = (Run * cap)()
        cnt = C.c_size_t(0)
        rc = self.lib.orc_oracle_rlev2_runs(s.ctypes.data, s.size, n, arr, cap, C.byref(cnt))
        return rc, [(r.kind, r.bit_width, r.length, r.byte_offset, r.byte_length)
                    for r in arr[:min(cnt.value, cap)]]

    def byte_rle(self, stream, n):
        s = _u8(stream)
        out = np.empty(n, dtype=np.uint8)
        rc = self.lib.orc_oracle_byte_rle(s.ctypes.data, s.size, n, out.ctypes.data, None)
        return rc, out

    def boolean(self, stream, n):
        s = _u8(stream)
        out = np.empty(n, dtype=np.uint8)
        ones = C.c_uint64(0)
        rc = self.lib.orc_oracle_boolean(s.ctypes.data, s.size, n, out.ctypes.data, C.byref(ones))
        return rc, out.astype(bool), ones.value

    def floats(self, stream, n, dtype):
        s = _u8(stream)
        out = np.empty(n, dtype=dtype)
        rc = self.lib.orc_oracle_float(s.ctypes.data, s.size, out.itemsize, n, out.ctypes.data)
        return rc, out

    def string_direct(self, length_stream, data, n, check_utf8=True):
        ls, d = _u8(length_stream), _u8(data)
        offsets = np.empty(n + 1, dtype=np.int64)
        out = np.empty(max(d.size, 1), dtype=np.uint8)
        rc = self.lib.orc_oracle_string_direct(ls.ctypes.data, ls.size, d.ctypes.data, d.size, n,
                                               int(check_utf8), offsets.ctypes.data,
                                               out.ctypes.data)
        if rc:
            return rc, offsets, out
        return rc, offsets, out[:offsets[n]]

    def string_dictionary(self, dict_offsets, dict_bytes, indices):
        do = np.ascontiguousarray(dict_offsets, dtype=np.int64)
        db = _u8(dict_bytes)
        idx = np.ascontiguousarray(indices, dtype=np.int64)
        n = idx.size
        offsets = np.empty(n + 1, dtype=np.int64)
        lens = np.diff(do)
        cap = int(lens.max()) * n if n and lens.size else 0
        out = np.empty(max(cap, 1), dtype=np.uint8)
        rc = self.lib.orc_oracle_string_dictionary(do.ctypes.data, db.ctypes.data, do.size - 1,
                                                   idx.ctypes.data, n, offsets.ctypes.data,
                                                   out.ctypes.data)
        if rc:
            return rc, offsets, out
        return rc, offsets, out[:offsets[n]]

    def zigzag(self, z):
        return self.lib.orc_oracle_zigzag(z)

    def unpack(self, data, bits, index):
        s = _u8(data)
        return self.lib.orc_oracle_unpack(s.ctypes.data, bits, index)


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liborc_oracle.so")
        src = os.path.join(ROOT, "oracle", "orc_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        _LIB = Oracle(C.CDLL(path))
    return _LIB


def strings_from(offsets, data):
    b = bytes(data)
    return [b[offsets[i]:offsets[i + 1]].decode("utf-8") for i in range(len(offsets) - 1)]
