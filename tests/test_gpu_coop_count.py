"""Columns of few long segments -- a file's entry point per row group, or none at all -- whose
headers are counted by a wavefront per segment: 64 byte offsets sized at once, the chain picked
out of them, only its offsets parsed in full.  The values are the oracle's (rle_v2.rs:135-175,
184-272, 283-301, 331-415); Delta headers with varints of every length, PatchedBase runs, runs
longer than the 64 offsets and segments too long for the wavefront's LDS slice included."""
import numpy as np
import pytest

import rle_encode as enc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from orc_format_amd import Context
    with Context(0) as c:
        yield c


def entries_every(starts, runs_per_entry):
    return np.array(starts[::runs_per_entry], dtype=np.uint64)


@pytest.mark.parametrize("signed", [False, True])
@pytest.mark.parametrize("kinds,max_len,runs_per_entry", [((0, 1), 6, 2000), ((0, 1, 2, 3), 40, 700),
                                                          ((0, 3), 12, 1500), ((1,), 3, 3000)])
def test_few_segments_of_short_runs(ctx, oracle, signed, kinds, max_len, runs_per_entry):
    rng = np.random.default_rng(1000 * len(kinds) + max_len + int(signed))
    stream, vals, starts = enc.random_stream(rng, 150 * runs_per_entry, signed, kinds=kinds,
                                             widths=[1, 2, 4, 7, 8, 13], max_len=max_len)
    entries = entries_every(starts, runs_per_entry)
    assert 64 <= entries.size <= 16_384
    out = ctx.rlev2(stream, vals.size, signed, entries)
    assert np.array_equal(out, vals)
    rc, ref, _ = oracle.rlev2(stream, vals.size, signed)
    assert rc == 0 and np.array_equal(out, ref)
    # the same stream without its entry points, and with fewer values wanted than it holds
    out = ctx.rlev2(stream, vals.size - 777, signed)
    assert np.array_equal(out, vals[:-777])


def delta_runs_with_long_varints(rng, n_runs, signed):
    """Delta runs whose base and first delta take 1 to 10 varint bytes each, between ShortRepeat
    runs: headers of 4 to 22 bytes, the longest of which do not fit a 16-byte window."""
    chunks, expect, starts, pos = [], [], [], 0
    for k in range(n_runs):
        if k % 3 == 2:
            v = int(rng.integers(0, 200))
            buf = enc.short_repeat(v, 5, signed)
            vals = np.full(5, v, dtype=np.int64)
        else:
            bits_b, bits_d = int(rng.integers(1, 64)), int(rng.integers(1, 63))
            base = int(rng.integers(0, 2**bits_b, dtype=np.uint64))
            if signed:
                base = (base >> 1) ^ -(base & 1)
            db = int(rng.integers(1, 2**bits_d, dtype=np.uint64)) * (-1 if rng.integers(0, 2) else 1)
            L = int(rng.integers(2, 9))
            buf = enc.delta(base, db, L - 2, 0, signed)
            with np.errstate(over="ignore"):
                vals = (np.uint64(base & (2**64 - 1)) +
                        np.arange(L, dtype=np.uint64) * np.uint64(db & (2**64 - 1))).view(np.int64)
        chunks.append(buf)
        expect.append(vals)
        starts.append(pos)
        pos += len(buf)
    return b"".join(chunks), np.concatenate(expect), starts


@pytest.mark.parametrize("signed", [False, True])
def test_delta_headers_with_varints_of_every_length(ctx, oracle, signed):
    rng = np.random.default_rng(77 + int(signed))
    stream, vals, starts = delta_runs_with_long_varints(rng, 120 * 400, signed)
    rc, ref, _ = oracle.rlev2(stream, vals.size, signed)
    assert rc == 0 and np.array_equal(ref, vals)
    out = ctx.rlev2(stream, vals.size, signed, entries_every(starts, 400))
    assert np.array_equal(out, vals)
    out = ctx.rlev2(stream, vals.size, signed)
    assert np.array_equal(out, vals)


def test_long_runs_between_short_ones_and_segments_too_long_for_lds(ctx, oracle):
    """Segments of a few kilobytes (walked out of LDS) next to segments of 60 KB of 512-value runs
    (walked from memory, by the stride of the run before), as the integer columns of a stripe are
    when they are decoded as one batch."""
    rng = np.random.default_rng(8)
    parts, vparts, entries, pos = [], [], [], 0
    for seg in range(200):
        entries.append(pos)
        if seg % 2 == 0:
            s, v, _ = enc.random_stream(rng, 1800, True, kinds=(0, 1), widths=[4], max_len=4)
        else:
            s, v, _ = enc.random_stream(rng, 20, True, kinds=(1,), widths=[48], max_len=512, min_len=512)
        parts.append(s)
        vparts.append(v)
        pos += len(s)
    stream, vals = b"".join(parts), np.concatenate(vparts)
    out = ctx.rlev2(stream, vals.size, True, np.array(entries, dtype=np.uint64))
    assert np.array_equal(out, vals)
    rc, ref, _ = oracle.rlev2(stream, vals.size, True)
    assert rc == 0 and np.array_equal(out, ref)


@pytest.mark.parametrize("n_runs", [1, 63, 64, 65, 500, 9000])
def test_one_segment(ctx, oracle, n_runs):
    """A stream without entry points and too short for any to be made up: one wavefront, out of LDS
    up to 32 KB and from memory beyond."""
    rng = np.random.default_rng(n_runs)
    stream, vals, _ = enc.random_stream(rng, n_runs, False, kinds=(0, 1, 2, 3), widths=[1, 3, 8, 17], max_len=20)
    out = ctx.rlev2(stream, vals.size, False)
    assert np.array_equal(out, vals)
    rc, ref, _ = oracle.rlev2(stream, vals.size, False)
    assert rc == 0 and np.array_equal(out, ref)


def test_errors_are_the_same(ctx, oracle):
    from orc_format_amd.error import OrcError
    rng = np.random.default_rng(9)
    stream, vals, starts = enc.random_stream(rng, 100 * 900, True, kinds=(0, 1, 3), widths=[4, 8], max_len=10)
    entries = entries_every(starts, 900)
    # an entry point that is not a header of the chain
    bad = entries.copy()
    bad[40] += 1
    with pytest.raises(OrcError):
        ctx.rlev2(stream, vals.size, True, bad)
    # the stream cut inside its last run, and inside a header: the oracle's verdict
    for cut_by in (1, len(stream) - starts[-1] - 1):
        cut = stream[:-cut_by]
        rc, _, _ = oracle.rlev2(cut, vals.size, True)
        assert rc != 0
        with pytest.raises(OrcError) as e:
            ctx.rlev2(cut, vals.size, True, entries)
        assert e.value.code == rc
    # more values wanted than the stream holds
    rc, _, _ = oracle.rlev2(stream, vals.size + 5, True)
    with pytest.raises(OrcError) as e:
        ctx.rlev2(stream, vals.size + 5, True, entries)
    assert e.value.code == rc
    # garbage after a good start: whatever the verdict, it is the oracle's
    junk = stream[:5000] + bytes(rng.integers(0, 256, size=4000, dtype=np.uint8))
    rc, ref, n_ok = oracle.rlev2(junk, vals.size, True)
    if rc == 0:
        assert np.array_equal(ctx.rlev2(junk, vals.size, True), ref)
    else:
        with pytest.raises(OrcError) as e:
            ctx.rlev2(junk, vals.size, True)
        assert e.value.code == rc
