"""The iterators that carry the names of `orc_format::read::decode` (src/read/decode/mod.rs:8-14)
beyond the RLE v2 ones: BooleanRleRunIter / BooleanRun, FloatIter, Values -- against the reference's
known-answer vectors (boolean_rle.rs:164-208), a host walk of the control bytes, and the C ABI's
orcd_byte_rle_runs directly."""
import numpy as np
import pytest

import rle_encode as enc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from orc_format_amd import default_context
    return default_context()


def walk(stream):
    """The runs of a byte-RLE stream, by the rule of boolean_rle.rs:40-62 (plain host loop)."""
    runs, pos = [], 0
    while pos < len(stream):
        c = stream[pos]
        if c < 128:
            runs.append((True, stream[pos + 1], c + 3))
            pos += 2
        else:
            n = 256 - c
            runs.append((False, bytes(stream[pos + 1:pos + 1 + n]), n))
            pos += 1 + n
    return runs


def test_boolean_run_kats(ctx):
    from orc_format_amd.read.decode import BooleanRleRunIter, BooleanRun
    # boolean_rle.rs:166: 0x61 announces 0x61 + 3 = 100 copies of the byte 0x00
    assert list(BooleanRleRunIter(bytes([0x61, 0x00]), ctx)) == [BooleanRun.Run(0, 100)]
    # boolean_rle.rs:178: 0xfe announces two literal bytes
    assert list(BooleanRleRunIter(bytes([0xFE, 0b01000100, 0b01000101]), ctx)) == \
        [BooleanRun.Literals(bytes([0b01000100, 0b01000101]))]
    # boolean_rle.rs:197
    assert list(BooleanRleRunIter(bytes([0xFF, 0x80]), ctx)) == [BooleanRun.Literals(b"\x80")]
    assert list(BooleanRleRunIter(b"", ctx)) == []


@pytest.mark.parametrize("seed,n", [(1, 50), (2, 5000), (3, 300_000)])
def test_boolean_runs_match_a_host_walk(ctx, seed, n):
    from orc_format_amd.read.decode import BooleanRleRunIter
    rng = np.random.default_rng(seed)
    # bursts of equal bytes between stretches of noise
    data = np.repeat(rng.integers(0, 256, size=n // 4 + 1, dtype=np.uint8),
                     rng.choice([1, 1, 2, 3, 7, 140], size=n // 4 + 1))[:n]
    stream = enc.byte_rle(data.tobytes())
    want = walk(stream)
    got = list(BooleanRleRunIter(stream, ctx))
    assert len(got) == len(want)
    for g, (is_run, payload, length) in zip(got, want):
        assert g.is_run == is_run
        assert g.payload == ((payload, length) if is_run else payload)


def test_byte_rle_runs_table(ctx):
    """The C entry point itself: positions, lengths, the number of runs that cover a prefix, and
    the error for a last run that is cut short."""
    from orc_format_amd.error import OrcError
    data = bytes([7] * 100 + [1, 2, 3, 4, 5] + [9] * 300 + [8, 8])
    stream = enc.byte_rle(data)
    table, total = ctx.byte_rle_runs(stream)
    assert total == len(data)
    out = ctx.byte_rle(stream, total)
    assert bytes(out) == data
    pos = 0
    for r, (is_run, payload, length) in zip(table, walk(stream)):
        assert int(r["is_run"]) == is_run and int(r["length"]) == length
        assert int(r["out_offset"]) == pos
        first = payload if is_run else payload[0]
        assert stream[int(r["payload_offset"])] == first
        pos += length
    assert pos == total
    # a prefix: the runs up to and including the one that holds byte 101
    part, held = ctx.byte_rle_runs(stream, 102)
    assert len(part) == 2 and held == 105
    # asking for more bytes than the stream holds
    with pytest.raises(OrcError) as e:
        ctx.byte_rle_runs(stream, len(data) + 1)
    assert e.value.code == 1
    # a literal group that announces more bytes than are left
    with pytest.raises(OrcError) as e:
        ctx.byte_rle_runs(stream + bytes([0xFD, 1]))
    assert e.value.code == 1


def test_boolean_runs_with_entry_points(ctx):
    from orc_format_amd.read.decode import BooleanRleRunIter
    rng = np.random.default_rng(4)
    chunks = [enc.byte_rle(rng.integers(0, 3, size=1000, dtype=np.uint8).tobytes()) for _ in range(40)]
    starts = np.cumsum([0] + [len(c) for c in chunks[:-1]]).astype(np.uint64)
    stream = b"".join(chunks)
    a = list(BooleanRleRunIter(stream, ctx))
    b = list(BooleanRleRunIter(stream, ctx, entry_points=starts))
    assert a == b and len(a) == len(walk(stream))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_float_iter(ctx, dtype):
    """FloatIter::<T, R>::new(reader, length) (float.rs:34-91): little-endian IEEE values, bit for
    bit (NaN payloads and signed zeros included); len(); DecodeFloat for a short stream."""
    from orc_format_amd.error import OrcError
    from orc_format_amd.read.decode import FloatIter
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2 ** (8 * np.dtype(dtype).itemsize) - 1, size=1000, dtype=np.uint64)
    raw = bits.astype("<u%d" % np.dtype(dtype).itemsize)
    raw[:4] = np.array([0.0, -0.0, np.inf, np.nan], dtype=dtype).view(raw.dtype)
    it = FloatIter(dtype, raw.tobytes(), raw.size, ctx)
    assert len(it) == raw.size
    got = it.to_numpy()
    assert got.dtype == np.dtype(dtype)
    assert (got.view(raw.dtype) == raw).all()
    it = FloatIter(dtype, raw.tobytes(), 3, ctx)
    assert [x for x in it][:2] == [0.0, -0.0]
    with pytest.raises(OrcError) as e:
        FloatIter(dtype, raw.tobytes()[:-1], raw.size, ctx).to_numpy()
    assert e.value.code == 5
    assert FloatIter(dtype, b"", 0, ctx).to_numpy().size == 0


def test_values():
    """Values::new(reader, scratch) / next(length) (variable_length.rs:5-28)."""
    from orc_format_amd.read.decode import Values
    v = Values(b"helloworld!", None)
    assert v.next(5) == b"hello"
    assert v.next(0) == b""
    assert v.next(5) == b"world"
    # a reader that runs dry hands out what it has, without an error (take + read_to_end)
    assert v.next(2) == b"!"
    assert v.next(2) == b""
