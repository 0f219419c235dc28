"""APIs to read from ORC: file metadata, stripe footers, column regions.

Mirrors `orc_format::read` (reference: src/read/mod.rs:38-159, src/read/column.rs:11-79) on top
of the C++ host reader inside liborcd.so.  No GPU is involved here; the streams a `Column` hands
out go to `read.decode`.
"""
import ctypes as C

import numpy as np

from .. import _native as N
from ..error import check
from ..proto import ColumnEncodingKind, CompressionKind, StreamKind
from . import decode  # noqa: F401


class FileMetadata:
    """read::FileMetadata (src/read/mod.rs:40-44): the parsed postscript and footer, plus the
    open file the later region reads use."""

    def __init__(self, handle, keepalive=None):
        self._lib = N.lib()
        self._h = handle
        self._keepalive = keepalive

    def close(self):
        if self._h:
            self._lib.orcd_file_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def compression(self):
        return CompressionKind(self._lib.orcd_file_compression(self._h))

    @property
    def number_of_rows(self):
        return self._lib.orcd_file_number_of_rows(self._h)

    @property
    def row_index_stride(self):
        return self._lib.orcd_file_row_index_stride(self._h)

    @property
    def stripes(self):
        out = []
        for i in range(self._lib.orcd_file_num_stripes(self._h)):
            si = N.StripeInfo()
            check(self._lib.orcd_file_stripe(self._h, i, C.byref(si)))
            out.append(si)
        return out

    @property
    def types(self):
        return [self._lib.orcd_file_type_kind(self._h, i)
                for i in range(self._lib.orcd_file_num_types(self._h))]


class StripeFooter:
    """proto::StripeFooter (src/proto.rs:157-164) with stream offsets resolved."""

    def __init__(self, handle):
        self._lib = N.lib()
        self._h = handle

    def __del__(self):
        try:
            if self._h:
                self._lib.orcd_stripe_footer_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def streams(self):
        out = []
        for i in range(self._lib.orcd_stripe_footer_num_streams(self._h)):
            si = N.StreamInfo()
            check(self._lib.orcd_stripe_footer_stream(self._h, i, C.byref(si)))
            out.append(si)
        return out


class Column:
    """read::Column (src/read/column.rs:11-79)."""

    def __init__(self, handle, metadata, footer):
        self._lib = N.lib()
        self._h = handle
        self._metadata = metadata  # keep the file and the footer alive
        self._footer = footer

    def __del__(self):
        try:
            if self._h:
                self._lib.orcd_column_free(self._h)
                self._h = None
        except Exception:
            pass

    def get_stream(self, kind):
        """Column::get_stream (column.rs:38-55): the stream's bytes (a numpy view into the
        column's region); raises InvalidKind when the column has no such stream."""
        ptr, ln = C.c_void_p(), C.c_size_t()
        check(self._lib.orcd_column_get_stream(self._h, int(kind), C.byref(ptr), C.byref(ln)))
        if ln.value == 0:
            return np.empty(0, dtype=np.uint8)
        buf = (C.c_uint8 * ln.value).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=np.uint8)
        arr.flags.writeable = False
        return _Owned(arr, self)

    def has_stream(self, kind):
        ptr, ln = C.c_void_p(), C.c_size_t()
        rc = self._lib.orcd_column_get_stream(self._h, int(kind), C.byref(ptr), C.byref(ln))
        if rc == N.INVALID_KIND:
            return False
        check(rc)
        return True

    def encoding(self):
        """Column::encoding().kind() (column.rs:58-60)."""
        kind, size = C.c_int(), C.c_uint32()
        check(self._lib.orcd_column_encoding(self._h, C.byref(kind), C.byref(size)))
        return ColumnEncodingKind(kind.value)

    def dictionary_size(self):
        """Column::dictionary_size() (column.rs:63-67)."""
        kind, size = C.c_int(), C.c_uint32()
        check(self._lib.orcd_column_encoding(self._h, C.byref(kind), C.byref(size)))
        return size.value

    def number_of_rows(self):
        """Column::number_of_rows() (column.rs:70-72)."""
        return self._lib.orcd_column_number_of_rows(self._h)

    def entry_points(self, kind):
        """Run-header offsets of stream `kind` from the column's ROW_INDEX (empty when the
        stripe has no usable index)."""
        ptr, n = C.c_void_p(), C.c_size_t()
        check(self._lib.orcd_column_entry_points(self._metadata._h, self._h, int(kind),
                                                 C.byref(ptr), C.byref(n)))
        if n.value == 0:
            return np.empty(0, dtype=np.uint64)
        buf = (C.c_uint64 * n.value).from_address(ptr.value)
        return np.frombuffer(buf, dtype=np.uint64).copy()

    def values_per_group(self, kind):
        """orcd_column_values_per_group: values of stream `kind` between two entry points when the
        index's values-to-skip are those of runs of 512 values throughout, else 0."""
        return int(self._lib.orcd_column_values_per_group(self._metadata._h, self._h, int(kind)))


class _Owned(np.ndarray):
    """numpy view that keeps the owning Column alive."""

    def __new__(cls, arr, owner):
        obj = arr.view(cls)
        obj._owner = owner
        return obj

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


def read_metadata(source):
    """read::read_metadata (src/read/mod.rs:46-89).  `source` is a path, or a bytes-like file
    image."""
    lib = N.lib()
    h = C.c_void_p()
    if isinstance(source, (str, bytes)) and not _looks_like_image(source):
        path = source.encode() if isinstance(source, str) else source
        check(lib.orcd_file_open(path, C.byref(h)))
        return FileMetadata(h)
    image = np.frombuffer(bytes(source), dtype=np.uint8)
    check(lib.orcd_file_open_memory(image.ctypes.data, image.size, C.byref(h)))
    return FileMetadata(h, keepalive=image)


def _looks_like_image(b):
    return isinstance(b, bytes) and b[:3] == b"ORC"


def read_stripe_footer(metadata, stripe):
    """read::read_stripe_footer (src/read/mod.rs:95-111)."""
    h = C.c_void_p()
    check(N.lib().orcd_read_stripe_footer(metadata._h, stripe, C.byref(h)))
    return StripeFooter(h)


def read_stripe_column(metadata, stripe, footer, column):
    """read::read_stripe_column (src/read/mod.rs:117-159)."""
    h = C.c_void_p()
    check(N.lib().orcd_read_stripe_column(metadata._h, stripe, footer._h, column, C.byref(h)))
    return Column(h, metadata, footer)
