"""Columns of many small segments -- more than 16 384 entry points a few hundred bytes apart --
whose headers are counted by wavefronts that hold 64 (or 32) segments' bytes in LDS.  The values are
the oracle's (rle_v2.rs:135-175, 184-272, 283-301, 331-415); entry points must still be headers on
the chain; a stretch of segments too long for LDS is walked from memory with the same result."""
import numpy as np
import pytest

import rle_encode as enc

pytestmark = pytest.mark.gpu

N_SEG = 17_000   # above the limit below which a wavefront per segment counts


@pytest.fixture()
def ctx():
    from orc_format_amd import Context
    with Context(0) as c:
        yield c


def segment_entries(starts, runs_per_entry):
    return np.array(starts[::runs_per_entry], dtype=np.uint64)


@pytest.mark.parametrize("signed", [False, True])
@pytest.mark.parametrize("kinds,max_len,runs_per_entry", [((0, 1), 64, 48), ((0, 1, 2, 3), 40, 24), ((0,), 10, 40)])
def test_small_segments_of_every_kind(ctx, oracle, signed, kinds, max_len, runs_per_entry):
    rng = np.random.default_rng(len(kinds) * 100 + max_len + int(signed))
    stream, vals, starts = enc.random_stream(rng, N_SEG * runs_per_entry, signed, kinds=kinds,
                                             widths=[1, 2, 4, 7, 8, 13], max_len=max_len)
    entries = segment_entries(starts, runs_per_entry)
    assert entries.size >= N_SEG
    out = ctx.rlev2(stream, vals.size, signed, entries)
    assert np.array_equal(out, vals)
    rc, ref, _ = oracle.rlev2(stream, vals.size, signed)
    assert rc == 0 and np.array_equal(out, ref)
    # fewer values than the stream holds: the tail is not decoded
    out = ctx.rlev2(stream, vals.size - 1234, signed, entries)
    assert np.array_equal(out, vals[:-1234])


def test_a_stretch_of_long_runs_among_them(ctx, oracle):
    """Here and there 20 runs of 512 wide values in a row: the segments around them span more
    than a wavefront's slice, which then walks from memory."""
    rng = np.random.default_rng(5)
    parts, vparts, starts, pos = [], [], [], 0
    for block in range(60):
        s, v, st = enc.random_stream(rng, 300 * 48, False, kinds=(0, 1), widths=[2, 4, 8], max_len=50)
        parts.append(s)
        vparts.append(v)
        starts += [pos + x for x in st]
        pos += len(s)
        if block % 7 == 3:
            s, v, st = enc.random_stream(rng, 20, False, kinds=(1,), widths=[56], max_len=512, min_len=512)
            parts.append(s)
            vparts.append(v)
            starts += [pos + x for x in st]
            pos += len(s)
    stream, vals = b"".join(parts), np.concatenate(vparts)
    entries = segment_entries(starts, 48)
    assert entries.size > 16_384
    out = ctx.rlev2(stream, vals.size, False, entries)
    assert np.array_equal(out, vals)


def test_errors_are_the_same(ctx, oracle):
    from orc_format_amd.error import OrcError
    rng = np.random.default_rng(6)
    stream, vals, starts = enc.random_stream(rng, N_SEG * 32, True, kinds=(0, 1), widths=[4, 8], max_len=30)
    entries = segment_entries(starts, 32)
    # an entry point that is not a header of the chain
    bad = entries.copy()
    bad[9000] += 1
    with pytest.raises(OrcError):
        ctx.rlev2(stream, vals.size, True, bad)
    # the stream cut inside its last run: the oracle's verdict
    cut = stream[:-1]
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
