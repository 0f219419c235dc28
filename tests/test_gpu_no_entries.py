"""Streams that come without entry points (no ROW_INDEX): the entry points that the decoders make
up for themselves -- guessed and then verified for RLE v2 (rle_v2.rs:480-514), computed exactly for
byte-RLE (boolean_rle.rs:40-62) -- must never change an answer or an error."""
import numpy as np
import pytest

import rle_encode as enc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from orc_format_amd import default_context
    return default_context()


# ---- RLE v2 --------------------------------------------------------------------------------------

@pytest.mark.parametrize("signed", [False, True])
@pytest.mark.parametrize("seed,n_runs,max_len,kinds", [
    (1, 30000, 12, (0, 1, 2, 3)),     # short runs of every kind
    (2, 60000, 6, (0, 1)),            # ShortRepeat / Direct only: the densest headers
    (3, 3000, 512, (0, 1, 2, 3)),     # long runs: the guess has little to land on
    (4, 20000, 40, (1, 3)),           # Direct / Delta
])
def test_rlev2_without_entries(ctx, oracle, signed, seed, n_runs, max_len, kinds):
    rng = np.random.default_rng(seed)
    stream, expect, starts = enc.random_stream(rng, n_runs, signed, kinds=kinds, max_len=max_len)
    assert len(stream) > 64 * 1024
    out = ctx.rlev2(stream, expect.size, signed)
    assert (out == expect).all()
    # ... and the same with a few real entry points, and for a prefix of the values
    some = np.array(starts[::997], dtype=np.uint64)
    assert (ctx.rlev2(stream, expect.size, signed, some) == expect).all()
    cut = expect.size // 2 + 3
    assert (ctx.rlev2(stream, cut, signed) == expect[:cut]).all()


def test_rlev2_without_entries_garbage_in_front_of_the_guesses(ctx, oracle):
    """Wide Direct runs of random bits between stretches of short runs: most offsets are no headers,
    and whatever the guesses were, the answer is the oracle's."""
    rng = np.random.default_rng(5)
    parts, vals = [], []
    for r in range(300):
        if r % 3 == 0:
            v = rng.integers(0, 2**63, size=512, dtype=np.uint64)
            parts.append(enc.direct(v, 64, False))
            vals.append(v.view(np.int64))
        else:
            s, e, _ = enc.random_stream(rng, 200, False, kinds=(0, 1), max_len=10)
            parts.append(s)
            vals.append(e)
    stream, expect = b"".join(parts), np.concatenate(vals)
    rc, ref, _ = oracle.rlev2(stream, expect.size, False)
    assert rc == 0 and (ref == expect).all()
    assert (ctx.rlev2(stream, expect.size, False) == expect).all()


def test_rlev2_without_entries_errors_stay_errors(ctx):
    from orc_format_amd.error import OrcError
    rng = np.random.default_rng(6)
    stream, expect, _ = enc.random_stream(rng, 30000, True, kinds=(0, 1, 3), max_len=12)
    with pytest.raises(OrcError) as e:
        ctx.rlev2(stream[:-3], expect.size, True)       # the last run is cut short
    assert e.value.code == 1
    with pytest.raises(OrcError) as e:
        ctx.rlev2(stream, expect.size + 1, True)        # more values than the stream holds
    assert e.value.code == 1
    assert (ctx.rlev2(stream[:-3], expect.size - 600, True) == expect[:-600]).all()


# ---- byte-RLE / booleans ---------------------------------------------------------------------------

def byte_streams():
    rng = np.random.default_rng(7)
    noise = rng.integers(0, 256, size=300_000, dtype=np.uint8)
    yield "literals", noise.tobytes()
    yield "all set", bytes([0xFF]) * 400_000                  # runs of 130: the period-2 stream 7f ff 7f ff ...
    yield "odd shift", bytes([1, 2, 3]) + bytes([0xFF]) * 400_000   # the same behind a literal group: odd offsets
    bursts = np.repeat(rng.integers(0, 256, size=60_000, dtype=np.uint8),
                       rng.choice([1, 1, 2, 3, 7, 140, 500], size=60_000))
    yield "bursts", bursts.tobytes()
    # bytes that look like control bytes inside literal groups
    yield "control-like", bytes(rng.choice([0x7F, 0xFF, 0x80, 0x00, 0x81], size=200_000).astype(np.uint8))


@pytest.mark.parametrize("name,data", list(byte_streams()), ids=[n for n, _ in byte_streams()])
def test_byte_rle_without_entries(ctx, oracle, name, data):
    for max_literal in (128, 5):
        stream = enc.byte_rle(data, max_literal=max_literal)
        if len(stream) < 20_000:
            stream = stream * (20_000 // len(stream) + 1)
            want = np.frombuffer(data * (20_000 // len(enc.byte_rle(data, max_literal=max_literal)) + 1), dtype=np.uint8)
        else:
            want = np.frombuffer(data, dtype=np.uint8)
        out = ctx.byte_rle(stream, want.size)
        assert (out == want).all()
        bits, ones = ctx.boolean(stream, want.size * 8 - 3)
        assert (bits == np.unpackbits(want)[:want.size * 8 - 3].astype(bool)).all()
        assert ones == int(np.unpackbits(want)[:want.size * 8 - 3].sum())
        table, total = ctx.byte_rle_runs(stream)
        assert total == want.size


def test_byte_rle_without_entries_errors_stay_errors(ctx):
    from orc_format_amd.error import OrcError
    rng = np.random.default_rng(8)
    data = rng.integers(0, 4, size=200_000, dtype=np.uint8).tobytes()
    stream = enc.byte_rle(data)
    with pytest.raises(OrcError) as e:
        ctx.byte_rle(stream, len(data) + 1)
    assert e.value.code == 1
    cut = stream[:-1]
    with pytest.raises(OrcError) as e:
        ctx.byte_rle(cut, len(data))
    assert e.value.code == 1
    assert bytes(ctx.byte_rle(cut, len(data) - 200)) == data[:-200]
