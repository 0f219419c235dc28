"""Stream decoders with the names of `orc_format::read::decode` (reference:
src/read/decode/mod.rs:8-14), backed by the HIP kernels through the C ABI.

The reference's iterators pull one value at a time from a `Read`; these take the bytes of one
stream (what `Column.get_stream` returns), decode the whole stream in one GPU call on first use,
and then iterate the result.  `reader` is any bytes-like object; `scratch` arguments are accepted
and ignored (the context recycles its own device buffers).
"""
import numpy as np

from ..context import default_context

SHORT_REPEAT, DIRECT, PATCHED_BASE, DELTA = 0, 1, 2, 3


class _Run:
    """One run of an RLE v2 stream: iterable over its values, with len()."""
    kind = None

    def __init__(self, values, bit_width):
        self._values = values
        self.bit_width = bit_width

    def __len__(self):
        return len(self._values)

    def __iter__(self):
        return iter(self._values.tolist())

    def to_numpy(self):
        return self._values


class ShortRepeat(_Run):
    kind = SHORT_REPEAT


class Direct(_Run):
    kind = DIRECT


class PatchedBase(_Run):
    kind = PATCHED_BASE


class Delta(_Run):
    kind = DELTA


_RUN_CLASSES = {SHORT_REPEAT: ShortRepeat, DIRECT: Direct, PATCHED_BASE: PatchedBase, DELTA: Delta}


class _RleV2RunIter:
    signed = False

    def __init__(self, reader, length, scratch=None, ctx=None, entry_points=None):
        self._reader = reader
        self._length = length
        self._ctx = ctx or default_context()
        self._entries = entry_points
        self._runs = None
        self._pos = 0

    def _decode(self):
        if self._runs is None:
            dtype = np.int64 if self.signed else np.uint64
            values = self._ctx.rlev2(self._reader, self._length, self.signed,
                                     self._entries).view(dtype)
            table = self._ctx.rlev2_runs(self._reader, self._length, self.signed, self._entries)
            self._runs = []
            for kind, width, length, out_off, _ in table:
                # like the reference, a last run may hold more values than were asked for
                self._runs.append(_RUN_CLASSES[kind](values[out_off:out_off + length], width))
            self._values = values

    def __iter__(self):
        return self

    def __next__(self):
        self._decode()
        if self._pos >= len(self._runs):
            raise StopIteration
        run = self._runs[self._pos]
        self._pos += 1
        return run

    def into_inner(self):
        return self._reader, None


class UnsignedRleV2RunIter(_RleV2RunIter):
    """UnsignedRleV2RunIter::new(reader, length, scratch) (rle_v2.rs:480-514)."""
    signed = False


class SignedRleV2RunIter(_RleV2RunIter):
    """SignedRleV2RunIter::new(reader, length, scratch) (rle_v2.rs:677-710)."""
    signed = True


class _RleV2Iter:
    signed = False

    def __init__(self, reader, length, scratch=None, ctx=None, entry_points=None):
        self._reader = reader
        self._length = length
        self._ctx = ctx or default_context()
        self._entries = entry_points
        self._it = None

    def to_numpy(self):
        dtype = np.int64 if self.signed else np.uint64
        return self._ctx.rlev2(self._reader, self._length, self.signed, self._entries).view(dtype)

    def __iter__(self):
        return self

    def __next__(self):
        if self._it is None:
            self._it = iter(self.to_numpy().tolist())
        return next(self._it)

    def into_inner(self):
        return self._reader, None


class UnsignedRleV2Iter(_RleV2Iter):
    """UnsignedRleV2Iter::new(reader, length, scratch) (rle_v2.rs:517-562)."""
    signed = False


class SignedRleV2Iter(_RleV2Iter):
    """SignedRleV2Iter::new(reader, length, scratch) (rle_v2.rs:713-758)."""
    signed = True


class BooleanIter:
    """BooleanIter::new(reader, length) (boolean_rle.rs:65-158)."""

    def __init__(self, reader, length, ctx=None, entry_points=None):
        self._reader = reader
        self._length = length
        self._ctx = ctx or default_context()
        self._entries = entry_points
        self._it = None

    def to_numpy(self):
        return self._ctx.boolean(self._reader, self._length, self._entries)[0]

    def __iter__(self):
        return self

    def __next__(self):
        if self._it is None:
            self._it = iter(self.to_numpy().tolist())
        return next(self._it)

    def into_inner(self):
        return self._reader


class BooleanRun:
    """BooleanRun (boolean_rle.rs:7-12): `Run(value, length)` -- is_run, payload = (byte, length)
    -- or `Literals(bytes)` -- payload = the literal bytes."""

    def __init__(self, is_run, payload):
        self.is_run = is_run
        self.payload = payload

    def __eq__(self, other):
        return isinstance(other, BooleanRun) and (self.is_run, self.payload) == (other.is_run, other.payload)

    def __repr__(self):
        return ("Run(%d, %d)" % self.payload) if self.is_run else "Literals(%r)" % (self.payload,)

    @staticmethod
    def Run(value, length):
        return BooleanRun(True, (value, length))

    @staticmethod
    def Literals(data):
        return BooleanRun(False, bytes(data))


class BooleanRleRunIter:
    """BooleanRleRunIter::new(reader) (boolean_rle.rs:14-63): the byte-RLE runs of the stream.
    Two GPU calls on first use, whatever the number of runs: the control-byte scan, whose run
    table says where every run starts in the decoded bytes, and the byte-RLE decode.  A stream
    whose last run is cut short raises OutOfSpec (an Err item there); unlike the reference's
    iterator, which yields an Err forever once the reader is empty, this one stops at the end."""

    def __init__(self, reader, ctx=None, entry_points=None):
        self._reader = reader
        self._ctx = ctx or default_context()
        self._entries = entry_points
        self._table = None
        self._pos = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self._table is None:
            self._table, total = self._ctx.byte_rle_runs(self._reader, None, self._entries)
            self._bytes = self._ctx.byte_rle(self._reader, total, self._entries) if total else b""
        if self._pos >= len(self._table):
            raise StopIteration
        r = self._table[self._pos]
        self._pos += 1
        off, n = int(r["out_offset"]), int(r["length"])
        if r["is_run"]:
            return BooleanRun.Run(int(self._bytes[off]), n)
        return BooleanRun.Literals(self._bytes[off:off + n])

    def into_inner(self):
        return self._reader


class FloatIter:
    """FloatIter::<T, R>::new(reader, length) (float.rs:34-91); `dtype` is np.float32/64."""

    def __init__(self, dtype, reader, length, ctx=None):
        self._dtype = np.dtype(dtype)
        self._reader = reader
        self._length = length
        self._ctx = ctx or default_context()
        self._it = None

    def __len__(self):
        return self._length

    def to_numpy(self):
        return self._ctx.floats(self._reader, self._length, self._dtype)

    def __iter__(self):
        return self

    def __next__(self):
        if self._it is None:
            self._it = iter(self.to_numpy().tolist())
        return next(self._it)

    def into_inner(self):
        return self._reader


class Values:
    """Values::new(reader, scratch) (variable_length.rs:5-28): `next(length)` returns the next
    `length` bytes of the stream -- fewer when the stream runs dry, without an error, as
    `take(length).read_to_end` does there (the string decoders check the total against the DATA
    stream's length instead)."""

    def __init__(self, reader, scratch=None):
        self._reader = memoryview(bytes(reader))
        self._pos = 0

    def next(self, length):
        out = self._reader[self._pos:self._pos + length]
        self._pos += len(out)
        return bytes(out)

    def into_inner(self):
        return None
