"""Decode context: one GPU, one HIP stream, recycled device buffers (orcd_ctx of include/orcd.h).

The batch methods take host buffers (bytes / numpy) or raw device pointers (ints, e.g.
`tensor.data_ptr()`), and call straight through the C ABI.
"""
import ctypes as C

import numpy as np

from . import _native as N
from .error import check


def _host_u8(buf):
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf.view(np.uint8) if buf.dtype != np.uint8 else buf)
    return np.frombuffer(bytes(buf), dtype=np.uint8)


def _entries(entry_points):
    if entry_points is None or len(entry_points) == 0:
        return None, 0
    e = np.ascontiguousarray(entry_points, dtype=np.uint64)
    return e, e.size


def _column_dict(cd):
    """Host arrays of an orcd_column_data as numpy copies."""
    def arr(ptr, count, dtype):
        if count == 0:
            return np.empty(0, dtype=dtype)
        buf = (C.c_uint8 * (count * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).copy()
    out = {"type_kind": cd.type_kind, "num_values": cd.num_values,
           "validity": arr(cd.validity, cd.num_rows, np.uint8).astype(bool)}
    if cd.values:
        # FLOAT / DOUBLE by type; BYTE (tinyint) is signed; BOOLEAN stays one byte 0 / 1
        dtype = {(5, 4): np.float32, (6, 8): np.float64, (1, 1): np.int8}.get(
            (cd.type_kind, cd.value_size), {8: np.int64, 1: np.uint8}.get(cd.value_size))
        out["values"] = arr(cd.values, cd.num_slots, dtype)
    if cd.offsets:
        out["offsets"] = arr(cd.offsets, cd.num_slots + 1, np.int64)
        out["bytes"] = arr(cd.bytes, cd.num_bytes, np.uint8) if cd.bytes else np.empty(0, np.uint8)
    return out


class DeviceColumn:
    """One decoded column of one stripe whose arrays stayed on the GPU (orcd_column_data with
    on_device set): raw device addresses and counts.  The memory goes back to the context that
    made it on free() -- call it (or drop the object) before that context is closed."""

    def __init__(self, lib, cd):
        self._lib = lib
        self._cd = N.ColumnData.from_buffer_copy(cd)
        self.type_kind = cd.type_kind
        self.num_rows, self.num_values, self.num_slots = cd.num_rows, cd.num_values, cd.num_slots
        self.validity_ptr, self.values_ptr, self.value_size = cd.validity or 0, cd.values or 0, cd.value_size
        self.offsets_ptr, self.bytes_ptr, self.num_bytes = cd.offsets or 0, cd.bytes or 0, cd.num_bytes

    def free(self):
        if self._cd is not None:
            self._lib.orcd_column_data_free(C.byref(self._cd))
            self._cd = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    def __init__(self, device=0):
        self._lib = N.lib()
        self._h = C.c_void_p()
        check(self._lib.orcd_ctx_create(device, C.byref(self._h)))
        self.caller_stream = 0
        self.device = device

    def close(self):
        if self._h:
            self._lib.orcd_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    def set_stream(self, hip_stream):
        check(self._lib.orcd_ctx_set_stream(self._h, C.c_void_p(hip_stream or 0)), self._h)
        self.caller_stream = hip_stream or 0  # 0: the context's own stream

    def synchronize(self):
        check(self._lib.orcd_ctx_synchronize(self._h), self._h)

    def copy_to_host(self, device_ptr, count, dtype):
        """`count` items of `dtype` from a device address (an array of a DeviceColumn) as numpy."""
        out = np.empty(count, dtype=dtype)
        check(self._lib.orcd_ctx_copy_to_host(self._h, out.ctypes.data, device_ptr, out.nbytes), self._h)
        return out

    def set_cache_limit(self, device_bytes, host_bytes):
        check(self._lib.orcd_ctx_set_cache_limit(self._h, device_bytes, host_bytes), self._h)

    def release_cached(self):
        """Returns the device memory kept for the next column calls to the runtime."""
        check(self._lib.orcd_ctx_release_cached(self._h), self._h)

    def last_elapsed_ms(self):
        return float(self._lib.orcd_ctx_last_elapsed_ms(self._h))

    def phase_ms(self, phase):
        return float(self._lib.orcd_ctx_phase_ms(self._h, phase))

    def last_path(self):
        """0: the last RLE v2 decode took the general path, 1: the strided single-launch one."""
        return int(self._lib.orcd_ctx_last_path(self._h))

    def last_value_kernel(self):
        """orcd_ctx_last_value_kernel: the kernel that expanded the last RLE v2 call's values."""
        return self._lib.orcd_ctx_last_value_kernel(self._h).decode()

    def last_stripe_ms(self):
        """orcd_ctx_last_stripe_ms: (feed_ms, decode_ms) of the last decode_stripe call."""
        feed, dec = C.c_float(), C.c_float()
        check(self._lib.orcd_ctx_last_stripe_ms(self._h, C.byref(feed), C.byref(dec)), self._h)
        return feed.value, dec.value

    # ---- host-buffer calls ------------------------------------------------------------------
    def rlev2(self, stream, num_values, signed, entry_points=None, values_per_group=0):
        """orcd_rlev2_decode, or with values_per_group orcd_rlev2_decode_groups: the entry points
        are one per row group of that many values (ROW_INDEX positions of a file)."""
        s = _host_u8(stream)
        e, ne = _entries(entry_points)
        out = np.empty(num_values, dtype=np.int64)
        if values_per_group:
            rc = self._lib.orcd_rlev2_decode_groups(self._h, s.ctypes.data, s.size, num_values,
                                                    e.ctypes.data if ne else None, ne, int(values_per_group),
                                                    out.ctypes.data, N.SIGNED if signed else 0)
        else:
            rc = self._lib.orcd_rlev2_decode(self._h, s.ctypes.data, s.size, num_values,
                                             e.ctypes.data if ne else None, ne, out.ctypes.data,
                                             N.SIGNED if signed else 0)
        check(rc, self._h)
        return out

    def rlev1(self, stream, num_values, signed, entry_points=None):
        """orcd_rlev1_decode: integer RLE version 1 (files written as ORC 0.11)."""
        s = _host_u8(stream)
        e, ne = _entries(entry_points)
        out = np.empty(num_values, dtype=np.int64)
        rc = self._lib.orcd_rlev1_decode(self._h, s.ctypes.data, s.size, num_values,
                                         e.ctypes.data if ne else None, ne, out.ctypes.data,
                                         N.SIGNED if signed else 0)
        check(rc, self._h)
        return out

    def inflate_chunks(self, src, chunks, dst_len):
        """orcd_inflate_chunks: `chunks` is a sequence of (src_offset, src_length, dst_offset,
        dst_capacity); returns (dst bytes as a uint8 array of dst_len, sizes)."""
        s = _host_u8(src)
        table = np.zeros(len(chunks), dtype=[("src_offset", "<u8"), ("dst_offset", "<u8"),
                                             ("src_length", "<u4"), ("dst_capacity", "<u4")])
        for k, (so, sl, do, dc) in enumerate(chunks):
            table[k] = (so, do, sl, dc)
        out = np.zeros(dst_len, dtype=np.uint8)
        sizes = np.zeros(len(chunks), dtype=np.uint32)
        rc = self._lib.orcd_inflate_chunks(self._h, s.ctypes.data, s.size, table.ctypes.data, len(chunks),
                                           out.ctypes.data, dst_len, sizes.ctypes.data, 0)
        check(rc, self._h)
        return out, sizes

    def rlev2_runs(self, stream, num_values, signed=False, entry_points=None):
        s = _host_u8(stream)
        e, ne = _entries(entry_points)
        cap = max(16, min(num_values, s.size))  # a run takes at least two stream bytes
        arr = (N.OrcdRun * cap)()
        cnt = C.c_size_t(0)
        rc = self._lib.orcd_rlev2_runs(self._h, s.ctypes.data, s.size, num_values,
                                       e.ctypes.data if ne else None, ne, arr, cap, C.byref(cnt),
                                       N.SIGNED if signed else 0)
        check(rc, self._h)
        return [(r.kind, r.bit_width, r.length, r.out_offset, r.payload_offset)
                for r in arr[:min(cap, cnt.value)]]

    def byte_rle_runs(self, stream, num_bytes=None, entry_points=None):
        """The runs of a byte-RLE stream as a structured array (payload_offset, out_offset, length,
        is_run) and the number of decoded bytes they hold; num_bytes=None: every run of the stream."""
        s = _host_u8(stream)
        e, ne = _entries(entry_points)
        cap = max(16, s.size)  # a run takes at least two stream bytes
        arr = np.zeros(cap, dtype=np.dtype([("payload_offset", "<u8"), ("out_offset", "<u8"),
                                            ("length", "<u4"), ("is_run", "<u4")]))
        cnt, total = C.c_size_t(0), C.c_uint64(0)
        rc = self._lib.orcd_byte_rle_runs(self._h, s.ctypes.data, s.size,
                                          N.WHOLE_STREAM if num_bytes is None else num_bytes,
                                          e.ctypes.data if ne else None, ne,
                                          C.cast(arr.ctypes.data, C.POINTER(N.OrcdByteRun)), cap,
                                          C.byref(cnt), C.byref(total), 0)
        check(rc, self._h)
        return arr[:min(cap, cnt.value)], int(total.value)

    def byte_rle(self, stream, num_bytes, entry_points=None):
        s = _host_u8(stream)
        e, ne = _entries(entry_points)
        out = np.empty(num_bytes, dtype=np.uint8)
        rc = self._lib.orcd_byte_rle_decode(self._h, s.ctypes.data, s.size, num_bytes,
                                            e.ctypes.data if ne else None, ne, out.ctypes.data, 0)
        check(rc, self._h)
        return out

    def boolean(self, stream, num_bits, entry_points=None, want_bitmap=False):
        s = _host_u8(stream)
        e, ne = _entries(entry_points)
        out = np.empty(num_bits, dtype=np.uint8)
        bitmap = np.empty((num_bits + 7) // 8, dtype=np.uint8) if want_bitmap else None
        ones = C.c_uint64(0)
        rc = self._lib.orcd_boolean_decode(self._h, s.ctypes.data, s.size, num_bits,
                                           e.ctypes.data if ne else None, ne, out.ctypes.data,
                                           bitmap.ctypes.data if want_bitmap else None,
                                           C.byref(ones), 0)
        check(rc, self._h)
        if want_bitmap:
            return out.astype(bool), ones.value, bitmap
        return out.astype(bool), ones.value

    def floats(self, stream, num_values, dtype):
        s = _host_u8(stream)
        out = np.empty(num_values, dtype=dtype)
        rc = self._lib.orcd_float_decode(self._h, s.ctypes.data, s.size, out.itemsize, num_values,
                                         out.ctypes.data, 0)
        check(rc, self._h)
        return out

    def string_direct(self, length_stream, data, num_values, check_utf8=True,
                      length_entry_points=None):
        ls, d = _host_u8(length_stream), _host_u8(data)
        e, ne = _entries(length_entry_points)
        offsets = np.empty(num_values + 1, dtype=np.int64)
        out = np.empty(max(d.size, 1), dtype=np.uint8)
        rc = self._lib.orcd_string_direct_decode(
            self._h, ls.ctypes.data, ls.size, e.ctypes.data if ne else None, ne, d.ctypes.data,
            d.size, num_values, offsets.ctypes.data, out.ctypes.data,
            N.CHECK_UTF8 if check_utf8 else 0)
        check(rc, self._h)
        return offsets, out[:offsets[num_values]]

    def string_dictionary(self, dict_offsets, dict_bytes, indices):
        do = np.ascontiguousarray(dict_offsets, dtype=np.int64)
        db = _host_u8(dict_bytes)
        idx = np.ascontiguousarray(indices, dtype=np.int64)
        n = idx.size
        offsets = np.empty(n + 1, dtype=np.int64)
        lens = np.diff(do)
        cap = int(lens.max()) * n if n and lens.size else 0
        out = np.empty(max(cap, 1), dtype=np.uint8)
        rc = self._lib.orcd_string_dictionary_decode(
            self._h, do.ctypes.data, db.ctypes.data, do.size - 1, idx.ctypes.data, n,
            offsets.ctypes.data, out.ctypes.data, out.size, 0)
        check(rc, self._h)
        return offsets, out[:offsets[n]]

    def scatter_valid(self, validity, dense, carry=False):
        """Dense non-null values -> one slot per row (zeros, or with `carry` the next value, at
        null rows)."""
        v = np.ascontiguousarray(validity).view(np.uint8)
        d = np.ascontiguousarray(dense)
        n_values = d.size - (1 if carry else 0)
        out = np.empty(v.size, dtype=d.dtype)
        rc = self._lib.orcd_scatter_valid(self._h, v.ctypes.data, v.size, d.ctypes.data, n_values,
                                          d.itemsize, out.ctypes.data, N.SCATTER_CARRY if carry else 0)
        check(rc, self._h)
        return out

    def decode_column(self, metadata, stripe, footer, column, check_utf8=True, row_aligned=False):
        """One column of one stripe through orcd_decode_column (host results).  Returns a dict:
        type_kind, validity (bool[num_rows]), and `values` (numpy) or `offsets` + `bytes`."""
        cd = N.ColumnData()
        flags = (N.CHECK_UTF8 if check_utf8 else 0) | (N.ROW_ALIGNED if row_aligned else 0)
        check(self._lib.orcd_decode_column(self._h, metadata._h, stripe, footer._h, column, flags,
                                           C.byref(cd)), self._h)
        try:
            return _column_dict(cd)
        finally:
            self._lib.orcd_column_data_free(C.byref(cd))

    def decode_stripe(self, metadata, stripe, footer, columns, check_utf8=True, row_aligned=False,
                      device=False):
        """Several columns of one stripe through orcd_decode_stripe: a list of the dicts
        decode_column returns (host results), or with device=True of DeviceColumn objects whose
        arrays stay on the GPU."""
        cols = (C.c_uint32 * len(columns))(*columns)
        cds = (N.ColumnData * len(columns))()
        flags = (N.CHECK_UTF8 if check_utf8 else 0) | (N.ROW_ALIGNED if row_aligned else 0) | \
            (N.OUT_DEVICE if device else 0)
        check(self._lib.orcd_decode_stripe(self._h, metadata._h, stripe, footer._h, cols, len(columns),
                                           flags, cds), self._h)
        if device:
            return [DeviceColumn(self._lib, cd) for cd in cds]
        try:
            return [_column_dict(cd) for cd in cds]
        finally:
            for cd in cds:
                self._lib.orcd_column_data_free(C.byref(cd))

    def stripe_prefetch(self, metadata, stripe, footer, columns):
        """orcd_stripe_prefetch: starts reading these columns of this stripe and copying them to the
        device in the background; the decode_stripe call for the same stripe and columns then finds
        the bytes there.  `metadata` and `footer` must be kept until that call (or the context's
        end): the caller holds on to them."""
        cols = (C.c_uint32 * len(columns))(*columns)
        check(self._lib.orcd_stripe_prefetch(self._h, metadata._h, stripe, footer._h, cols, len(columns)),
              self._h)

    def rlev2_batch(self, buffer, slices, signed, out_ptr=0, device=False):
        """orcd_rlev2_decode_batch.  `slices`: (offset, length, num_values, entry_points or None)
        per stream inside `buffer` (host bytes / numpy, or with device=True a raw device address
        and its length as a tuple).  Returns the concatenated values (host), or with device=True
        writes them at device address out_ptr."""
        arr = (N.Rlev2Slice * len(slices))()
        keep = []
        total = 0
        for i, (off, ln, nv, ent) in enumerate(slices):
            e, ne = _entries(ent)
            keep.append(e)
            arr[i] = N.Rlev2Slice(off, ln, nv, e.ctypes.data if ne else None, ne)
            total += nv
        bad = C.c_size_t(0)
        if device:
            ptr, blen = buffer
            rc = self._lib.orcd_rlev2_decode_batch(self._h, ptr, blen, arr, len(slices), out_ptr or None,
                                                   C.byref(bad), (N.SIGNED if signed else 0) | N.IN_DEVICE | N.OUT_DEVICE)
            self.bad_slice = bad.value
            check(rc, self._h)
            return None
        b = _host_u8(buffer)
        out = np.empty(total, dtype=np.int64)
        rc = self._lib.orcd_rlev2_decode_batch(self._h, b.ctypes.data, b.size, arr, len(slices),
                                               out.ctypes.data, C.byref(bad), N.SIGNED if signed else 0)
        self.bad_slice = bad.value
        check(rc, self._h)
        return out

    def boolean_batch(self, buffer, slices):
        """orcd_boolean_decode_batch on host buffers.  `slices`: (offset, length, num_bits,
        entry_points or None).  Returns (list of bool arrays, one per slice; list of set-bit counts)."""
        arr = (N.Rlev2Slice * len(slices))()
        keep = []
        starts, pos = [], 0
        for i, (off, ln, nv, ent) in enumerate(slices):
            e, ne = _entries(ent)
            keep.append(e)
            arr[i] = N.Rlev2Slice(off, ln, nv, e.ctypes.data if ne else None, ne)
            starts.append(pos)
            pos += 8 * ((nv + 7) // 8)
        b = _host_u8(buffer)
        out = np.empty(max(pos, 1), dtype=np.uint8)
        counts = (C.c_uint64 * len(slices))()
        bad = C.c_size_t(0)
        rc = self._lib.orcd_boolean_decode_batch(self._h, b.ctypes.data, b.size, arr, len(slices),
                                                 out.ctypes.data, counts, C.byref(bad), 0)
        self.bad_slice = bad.value
        check(rc, self._h)
        return ([out[s:s + sl[2]].astype(bool) for s, sl in zip(starts, slices)], list(counts))

    # ---- device-pointer calls (raw addresses) -------------------------------------------------
    def rlev2_device(self, stream_ptr, length, num_values, signed, out_ptr, entries_ptr=0,
                     n_entries=0, values_per_group=0):
        flags = (N.SIGNED if signed else 0) | N.IN_DEVICE | N.OUT_DEVICE
        if values_per_group:
            rc = self._lib.orcd_rlev2_decode_groups(self._h, stream_ptr, length, num_values,
                                                    entries_ptr or None, n_entries, int(values_per_group),
                                                    out_ptr, flags)
        else:
            rc = self._lib.orcd_rlev2_decode(self._h, stream_ptr, length, num_values,
                                             entries_ptr or None, n_entries, out_ptr, flags)
        check(rc, self._h)

    def rlev2_device_noalloc(self, stream_ptr, length, num_values, signed, out_ptr, entries_ptr=0,
                             n_entries=0):
        """orcd_rlev2_decode_noalloc: the same decode with every table in the library's static
        device array; device pointers only, limits instead of growth (ORCD_INVALID_ARGUMENT)."""
        rc = self._lib.orcd_rlev2_decode_noalloc(self._h, stream_ptr, length, num_values,
                                                 entries_ptr or None, n_entries, out_ptr,
                                                 N.SIGNED if signed else 0)
        check(rc, self._h)

    def boolean_device(self, stream_ptr, length, num_bits, out_bytes_ptr=0, out_bitmap_ptr=0,
                       entries_ptr=0, n_entries=0, want_count=True):
        ones = C.c_uint64(0)
        rc = self._lib.orcd_boolean_decode(self._h, stream_ptr, length, num_bits,
                                           entries_ptr or None, n_entries, out_bytes_ptr or None,
                                           out_bitmap_ptr or None,
                                           C.byref(ones) if want_count else None,
                                           N.IN_DEVICE | N.OUT_DEVICE)
        check(rc, self._h)
        return ones.value

    def float_device(self, stream_ptr, length, elem_size, num_values, out_ptr):
        rc = self._lib.orcd_float_decode(self._h, stream_ptr, length, elem_size, num_values,
                                         out_ptr, N.IN_DEVICE | N.OUT_DEVICE)
        check(rc, self._h)

    def string_direct_device(self, length_ptr, length_len, data_ptr, data_len, num_values,
                             offsets_ptr, bytes_out_ptr=0, entries_ptr=0, n_entries=0,
                             check_utf8=False):
        rc = self._lib.orcd_string_direct_decode(
            self._h, length_ptr, length_len, entries_ptr or None, n_entries, data_ptr, data_len,
            num_values, offsets_ptr, bytes_out_ptr or None,
            N.IN_DEVICE | N.OUT_DEVICE | (N.CHECK_UTF8 if check_utf8 else 0))
        check(rc, self._h)


_default = {}


def default_context(device=0):
    """Process-wide context per device, created on first use."""
    ctx = _default.get(device)
    if ctx is None:
        ctx = _default[device] = Context(device)
    return ctx
