"""The single-launch decode of regular columns (every 512 values one Direct run, equal-sized runs
between entry points: ORCD_PATH_STRIDED) against the oracle, and its fall-back to the general path
on every column that is not like that -- the answers must be the same either way
(rle_v2.rs:135-175, 480-514)."""
import numpy as np
import pytest

import rle_encode as enc

pytestmark = pytest.mark.gpu

PATH_GENERAL, PATH_STRIDED = 0, 1


@pytest.fixture()
def ctx():
    # a context of its own: the back-off after failed attempts is per context
    from orc_format_amd import Context
    with Context(0) as c:
        yield c


def direct_column(rng, n_runs, width, signed, last_len=512, runs_per_entry=20):
    """n_runs Direct runs of 512 values (the last of last_len), entry points every runs_per_entry."""
    parts, starts, vals = [], [], []
    pos = 0
    for r in range(n_runs):
        n = last_len if r == n_runs - 1 else 512
        hi = 1 << (width - (1 if signed else 0))
        if signed:
            v = rng.integers(-(hi // 2) if width > 1 else 0, max(1, hi // 2), size=n, dtype=np.int64)
        else:
            v = rng.integers(0, hi, size=n, dtype=np.uint64) if width < 64 else \
                rng.integers(0, 2**63, size=n, dtype=np.uint64) * 2 + 1
        b = enc.direct(v, width, signed)
        starts.append(pos)
        pos += len(b)
        parts.append(b)
        vals.append(np.asarray(v).astype(np.int64) if signed else np.asarray(v, dtype=np.uint64).view(np.int64))
    entries = np.array(starts[::runs_per_entry], dtype=np.uint64) if runs_per_entry else None
    return b"".join(parts), np.concatenate(vals), entries, starts


def check(ctx, oracle, stream, n, signed, entries, want_path, expect=None):
    out = ctx.rlev2(stream, n, signed, entries)
    rc, ref, _ = oracle.rlev2(stream, n, signed)
    assert rc == 0
    assert (out == ref).all()
    if expect is not None:
        assert (out == expect[:n]).all()
    if want_path is not None:
        assert ctx.last_path() == want_path


@pytest.mark.parametrize("width", [1, 3, 8, 17, 24, 26, 32, 40, 64])
@pytest.mark.parametrize("signed", [False, True])
def test_strided_every_width(ctx, oracle, width, signed):
    rng = np.random.default_rng(width * 2 + signed)
    stream, vals, entries, _ = direct_column(rng, 47, width, signed)
    check(ctx, oracle, stream, vals.size, signed, entries, PATH_STRIDED, vals)


@pytest.mark.parametrize("last_len,cut", [(512, 0), (512, 37), (1, 0), (300, 0), (300, 123)])
def test_strided_last_window(ctx, oracle, last_len, cut):
    """The last run may be short, and the values wanted may end inside it."""
    rng = np.random.default_rng(last_len + cut)
    stream, vals, entries, _ = direct_column(rng, 61, 17, True, last_len=last_len)
    check(ctx, oracle, stream, vals.size - cut, True, entries, PATH_STRIDED, vals)


def test_strided_without_entry_points(ctx, oracle):
    rng = np.random.default_rng(5)
    stream, vals, _, _ = direct_column(rng, 33, 11, False, runs_per_entry=0)
    check(ctx, oracle, stream, vals.size, False, None, PATH_STRIDED, vals)
    # one entry point (the start of the stream) is the same thing
    check(ctx, oracle, stream, vals.size, False, np.zeros(1, dtype=np.uint64), PATH_STRIDED, vals)


def test_strided_last_run_of_another_kind(ctx, oracle):
    """The run that ends the column is decoded by the general single-run code: any kind will do."""
    rng = np.random.default_rng(6)
    stream, vals, entries, _ = direct_column(rng, 40, 9, True)
    tail = enc.delta(1000, 7, 98, 0, True)   # 100 values: 1000, 1007, ...
    tail_vals = 1000 + 7 * np.arange(100, dtype=np.int64)
    check(ctx, oracle, stream + tail, vals.size + 100, True, entries, PATH_STRIDED,
          np.concatenate([vals, tail_vals]))


def test_one_run_breaks_the_stride(ctx, oracle):
    """A column that is regular but for one run in the middle -- another width, hence another size --
    is not for the strided path: the plan-and-expand path takes it when there are entry points to
    plan between, the general path otherwise, with the same answer."""
    rng = np.random.default_rng(7)
    parts, vals, starts, pos = [], [], [], 0
    for r in range(60):
        w = 12 if r == 31 else 17
        v = rng.integers(-(1 << (w - 2)), 1 << (w - 2), size=512, dtype=np.int64)
        b = enc.direct(v, w, True)
        starts.append(pos)
        pos += len(b)
        parts.append(b)
        vals.append(v)
    stream, vals = b"".join(parts), np.concatenate(vals)
    check(ctx, oracle, stream, vals.size, True, np.array(starts[::20], dtype=np.uint64), 2, vals)
    check(ctx, oracle, stream, vals.size, True, None, PATH_GENERAL, vals)


def test_a_short_run_in_the_middle(ctx, oracle):
    """Equal sizes are not enough: a run of fewer values shifts every later window."""
    rng = np.random.default_rng(8)
    parts, vals, starts, pos = [], [], [], 0
    for r in range(50):
        n = 256 if r == 20 else 512
        w = 16 if r == 20 else 8     # 2 + 256 * 16 / 8 = 2 + 512 * 8 / 8: the same size in bytes
        v = rng.integers(0, 1 << w, size=n, dtype=np.uint64)
        b = enc.direct(v, w, False)
        assert len(b) == 514
        starts.append(pos)
        pos += len(b)
        parts.append(b)
        vals.append(v.view(np.int64))
    stream, vals = b"".join(parts), np.concatenate(vals)
    check(ctx, oracle, stream, vals.size, False, np.array(starts[::10], dtype=np.uint64), PATH_GENERAL, vals)


def test_delta_column_is_not_for_the_strided_path(ctx, oracle):
    """Full Delta runs have no common size: the strided attempt fails and the plan-and-expand path
    (tests/test_gpu_planned.py) takes the column."""
    rng = np.random.default_rng(9)
    parts, vals, starts, pos, t = [], [], [], 0, 1_600_000_000_000
    for r in range(40):
        d = rng.integers(1, 1000, size=511, dtype=np.int64)
        v = t + np.concatenate([[0], np.cumsum(d)])
        b = enc.delta(int(v[0]), int(d[0]), d[1:], 10, True)
        t = int(v[-1]) + 1
        starts.append(pos)
        pos += len(b)
        parts.append(b)
        vals.append(v)
    stream, vals = b"".join(parts), np.concatenate(vals)
    check(ctx, oracle, stream, vals.size, True, np.array(starts[::20], dtype=np.uint64), 2, vals)


def test_entry_points_that_are_not_evenly_spaced(ctx, oracle):
    """Entry points every 19 or 20 runs, as the ROW_INDEX of a file with a stride of 10 000 rows
    gives them: not what the strided path assumes, so the general path answers."""
    rng = np.random.default_rng(10)
    stream, vals, _, starts = direct_column(rng, 100, 17, True, runs_per_entry=0)
    picks = [0, 19, 39, 58, 78, 97]
    check(ctx, oracle, stream, vals.size, True, np.array([starts[i] for i in picks], dtype=np.uint64),
          PATH_GENERAL, vals)


def test_wrong_entry_point_is_still_an_error(ctx):
    """An entry point that is not a run boundary stays OutOfSpec (the strided attempt notices that
    the runs do not fit between the entries and leaves the verdict to the header scan)."""
    from orc_format_amd.error import OrcError
    rng = np.random.default_rng(11)
    stream, vals, entries, _ = direct_column(rng, 60, 17, True)
    bad = entries.copy()
    bad[1] += 20
    with pytest.raises(OrcError) as e:
        ctx.rlev2(stream, vals.size, True, bad)
    assert e.value.code == 1
    assert ctx.last_path() == PATH_GENERAL


def test_truncated_stream_is_an_error(ctx):
    from orc_format_amd.error import OrcError
    rng = np.random.default_rng(12)
    stream, vals, entries, _ = direct_column(rng, 60, 17, True)
    with pytest.raises(OrcError) as e:
        ctx.rlev2(stream[:-5], vals.size, True, entries)
    assert e.value.code == 1


def test_back_off_and_recovery(ctx, oracle):
    """Failed attempts make the context skip a few, after which regular columns are fast again."""
    rng = np.random.default_rng(13)
    irregular, expect, starts = enc.random_stream(rng, 3000, True)
    regular, vals, entries, _ = direct_column(rng, 30, 17, True)
    for _ in range(4):
        check(ctx, oracle, irregular, expect.size, True, None, PATH_GENERAL)
    seen = []
    for _ in range(20):
        check(ctx, oracle, regular, vals.size, True, entries, None, vals)
        seen.append(ctx.last_path())
    assert seen[-1] == PATH_STRIDED and PATH_GENERAL in seen


def test_device_pointers_full_window_alignment(ctx, oracle):
    """Device-pointer call; an output that is not 16-byte aligned is not for the strided kernel
    (the plan-and-expand path, which has no such condition, takes the column)."""
    import torch
    rng = np.random.default_rng(14)
    stream, vals, entries, _ = direct_column(rng, 64, 17, True)
    d_stream = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy()).cuda()
    d_entries = torch.from_numpy(entries.view(np.int64).copy()).cuda()
    buf = torch.zeros(vals.size + 2, dtype=torch.int64, device="cuda")
    for shift, want in ((0, PATH_STRIDED), (1, 2)):
        out = buf[shift:shift + vals.size]
        out.zero_()
        ctx.rlev2_device(d_stream.data_ptr(), d_stream.numel(), vals.size, True, out.data_ptr(),
                         d_entries.data_ptr(), d_entries.numel())
        ctx.synchronize()
        assert (out.cpu().numpy() == vals).all()
        assert ctx.last_path() == want
