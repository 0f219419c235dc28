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
    """BooleanRun (boolean_rle.rs:9-12): `Run(value, length)` or `Literals(bytes)`."""

    def __init__(self, is_run, payload):
        self.is_run = is_run
        self.payload = payload


class BooleanRleRunIter:
    """BooleanRleRunIter::new(reader) (boolean_rle.rs:14-63): the byte-RLE runs of the stream.
    The bytes come from the GPU byte-RLE decoder; run boundaries are re-read from the control
    bytes on the host only to label them."""

    def __init__(self, reader, ctx=None):
        self._reader = bytes(reader)
        self._ctx = ctx or default_context()
        self._pos = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self._pos >= len(self._reader):
            raise StopIteration
        control = self._reader[self._pos]
        if control < 128:
            n, used = control + 3, 2
        else:
            n, used = 256 - control, 1 + 256 - control
        chunk = self._reader[self._pos:self._pos + used]
        data = self._ctx.byte_rle(chunk, n)
        self._pos += used
        if control < 128:
            return BooleanRun(True, (int(data[0]), n))
        return BooleanRun(False, bytes(data))

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
    `length` bytes of the stream."""

    def __init__(self, reader, scratch=None):
        self._reader = memoryview(bytes(reader))
        self._pos = 0

    def next(self, length):
        from ..error import OrcError
        from .. import _native as N
        if length > len(self._reader) - self._pos:
            raise OrcError(N.OUT_OF_SPEC)
        out = self._reader[self._pos:self._pos + length]
        self._pos += length
        return bytes(out)

    def into_inner(self):
        return None
