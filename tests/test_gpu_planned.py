"""Columns of full runs of any kind -- every 512 values one run, equally many runs between entry
points -- decoded by the plan-and-expand path (ORCD_PATH_PLANNED: run headers guessed by
interpolation between entry points and proven by their links) against the oracle, and the
fall-back to the general path for everything that is not like that: the answers must be the same
either way (rle_v2.rs:184-272, 331-415, 441-461, 480-514)."""
import numpy as np
import pytest

import rle_encode as enc

pytestmark = pytest.mark.gpu

PATH_GENERAL, PATH_STRIDED, PATH_PLANNED = 0, 1, 2


@pytest.fixture()
def ctx():
    # a context of its own: which speculative path is tried first, and the back-off, are per context
    from orc_format_amd import Context
    with Context(0) as c:
        yield c


def delta_column(rng, n_runs, width, signed, descending=False, last_len=512, small_steps=0.0):
    """n_runs full Delta runs (the last of last_len values) of a monotonic column.  small_steps: share
    of runs whose first step is below 64, so that its varint -- and the run -- is a byte shorter."""
    parts, starts, vals = [], [], []
    pos, t = 0, (1_600_000_000_000 if signed or not descending else 2**62)
    sign = -1 if descending else 1
    for r in range(n_runs):
        n = last_len if r == n_runs - 1 else 512
        n = max(n, 2)
        d = rng.integers(0, 1 << width, size=n - 1, dtype=np.uint64).astype(np.int64) if width < 63 else \
            rng.integers(0, 2**62, size=n - 1, dtype=np.int64)
        d[0] = int(rng.integers(1, 64)) if rng.random() < small_steps else int(rng.integers(64, 1000))
        v = t + sign * np.concatenate([[0], np.cumsum(d)])
        b = enc.delta(int(v[0]), sign * int(d[0]), d[1:].astype(np.uint64), width, signed)
        t = int(v[-1]) + sign
        starts.append(pos)
        pos += len(b)
        parts.append(b)
        vals.append(v.astype(np.int64))
    return b"".join(parts), np.concatenate(vals), starts


def check(ctx, oracle, stream, n, signed, entries, want_path, expect=None):
    out = ctx.rlev2(stream, n, signed, entries)
    rc, ref, _ = oracle.rlev2(stream, n, signed)
    assert rc == 0
    assert (out == ref).all()
    if expect is not None:
        assert (out == expect[:n]).all()
    if want_path is not None:
        assert ctx.last_path() == want_path


def every(starts, k):
    return np.array(starts[::k], dtype=np.uint64)


@pytest.mark.parametrize("width", [2, 5, 10, 16, 24, 32, 40, 48])
@pytest.mark.parametrize("signed", [False, True])
def test_planned_delta_widths(ctx, oracle, width, signed):
    rng = np.random.default_rng(width * 2 + signed)
    stream, vals, starts = delta_column(rng, 47, width, signed, small_steps=0.3)
    check(ctx, oracle, stream, vals.size, signed, every(starts, 20), PATH_PLANNED, vals)


def test_planned_descending(ctx, oracle):
    rng = np.random.default_rng(3)
    stream, vals, starts = delta_column(rng, 83, 10, True, descending=True, small_steps=0.5)
    check(ctx, oracle, stream, vals.size, True, every(starts, 20), PATH_PLANNED, vals)


@pytest.mark.parametrize("runs_per_entry,n_runs", [(1, 40), (2, 41), (7, 50), (20, 200), (20, 201), (64, 300), (100, 350)])
def test_planned_segment_shapes(ctx, oracle, runs_per_entry, n_runs):
    """Few and many runs per segment; a last segment of one run and of as many as the others."""
    rng = np.random.default_rng(runs_per_entry * 1000 + n_runs)
    stream, vals, starts = delta_column(rng, n_runs, 10, True, small_steps=0.2)
    entries = every(starts, runs_per_entry)
    if entries.size < 2:
        pytest.skip("one segment: nothing to interpolate between")
    check(ctx, oracle, stream, vals.size, True, entries, PATH_PLANNED, vals)


@pytest.mark.parametrize("last_len,cut", [(512, 0), (512, 37), (2, 0), (300, 0), (300, 123)])
def test_planned_last_window(ctx, oracle, last_len, cut):
    """The last run may be short, and the values wanted may end inside it."""
    rng = np.random.default_rng(last_len + cut)
    stream, vals, starts = delta_column(rng, 61, 16, True, last_len=last_len)
    check(ctx, oracle, stream, vals.size - cut, True, every(starts, 20), PATH_PLANNED, vals)


def test_planned_mix_of_kinds(ctx, oracle):
    """Full runs of every kind and width: sizes from 4 bytes to 4 KiB, so most guesses miss and the
    lanes walk their segment instead -- the links hold all the same."""
    rng = np.random.default_rng(21)
    for signed in (False, True):
        stream, vals, starts = enc.random_stream(rng, 150, signed, kinds=(1, 2, 3), min_len=512, max_len=512)
        check(ctx, oracle, stream, vals.size, signed, every(starts, 10), PATH_PLANNED, vals)


def test_planned_fixed_delta_runs(ctx, oracle):
    """Runs of a fixed step are a handful of bytes each: several headers in every search window."""
    parts, starts, vals, pos = [], [], [], 0
    for r in range(200):
        b = enc.delta(1000 * r, 3, 510, 0, True)
        starts.append(pos)
        pos += len(b)
        parts.append(b)
        vals.append(1000 * r + 3 * np.arange(512, dtype=np.int64))
    stream, vals = b"".join(parts), np.concatenate(vals)
    check(ctx, oracle, stream, vals.size, True, every(starts, 20), PATH_PLANNED, vals)


def test_header_bytes_inside_the_payload(ctx, oracle):
    """Payload bytes that look like the run header, right where the guesses land: more than one hit
    sends the lane on the walk, and a lone false hit is caught by the links."""
    rng = np.random.default_rng(22)
    width = 16
    h = bytes([(3 << 6) | (enc.width_code(width) << 1) | 1, 0xFF])
    fake = int.from_bytes(h, "big")
    parts, starts, vals, pos, t = [], [], [], 0, 10**9
    for r in range(80):
        d = rng.integers(1, 1 << width, size=511, dtype=np.int64)
        d[0] = int(rng.integers(64, 1000))
        # the last deltas of the run, just in front of the next header, repeat its two bytes
        if r % 3 == 0:
            d[-8:] = fake
        if r % 3 == 1:
            d[1:6] = fake   # ... and the first ones, just behind this run's header
        v = t + np.concatenate([[0], np.cumsum(d)])
        b = enc.delta(int(v[0]), int(d[0]), d[1:].astype(np.uint64), width, True)
        t = int(v[-1]) + 1
        starts.append(pos)
        pos += len(b)
        parts.append(b)
        vals.append(v.astype(np.int64))
    stream, vals = b"".join(parts), np.concatenate(vals)
    check(ctx, oracle, stream, vals.size, True, every(starts, 20), PATH_PLANNED, vals)


def test_one_short_run_breaks_the_plan(ctx, oracle):
    """One run of fewer values in the middle shifts every later window: general path, same answer."""
    rng = np.random.default_rng(23)
    parts, starts, vals, pos, t = [], [], [], 0, 5 * 10**11
    for r in range(60):
        n = 400 if r == 31 else 512
        d = rng.integers(1, 1000, size=n - 1, dtype=np.int64)
        d[0] = max(int(d[0]), 64)
        v = t + np.concatenate([[0], np.cumsum(d)])
        b = enc.delta(int(v[0]), int(d[0]), d[1:].astype(np.uint64), 10, True)
        t = int(v[-1]) + 1
        starts.append(pos)
        pos += len(b)
        parts.append(b)
        vals.append(v.astype(np.int64))
    stream, vals = b"".join(parts), np.concatenate(vals)
    check(ctx, oracle, stream, vals.size, True, every(starts, 20), PATH_GENERAL, vals)


def test_uneven_entry_points_take_the_general_path(ctx, oracle):
    rng = np.random.default_rng(24)
    stream, vals, starts = delta_column(rng, 100, 10, True)
    picks = [0, 19, 39, 58, 78, 97]
    check(ctx, oracle, stream, vals.size, True, np.array([starts[i] for i in picks], dtype=np.uint64),
          PATH_GENERAL, vals)


def test_wrong_entry_point_is_still_an_error(ctx):
    from orc_format_amd.error import OrcError
    rng = np.random.default_rng(25)
    stream, vals, starts = delta_column(rng, 60, 10, True)
    bad = every(starts, 20)
    bad[1] += 3
    with pytest.raises(OrcError) as e:
        ctx.rlev2(stream, vals.size, True, bad)
    assert e.value.code == 1
    assert ctx.last_path() == PATH_GENERAL


def test_truncated_stream_is_an_error(ctx):
    from orc_format_amd.error import OrcError
    rng = np.random.default_rng(26)
    stream, vals, starts = delta_column(rng, 60, 10, True)
    with pytest.raises(OrcError) as e:
        ctx.rlev2(stream[:-5], vals.size, True, every(starts, 20))
    assert e.value.code == 1


def test_the_path_that_worked_is_tried_first(ctx, oracle):
    """Delta columns after a Direct column and back: each kind of regular column finds its fast path
    again, whatever the context decoded before (a plan that meets nothing but equal-sized Direct
    runs hands the next column to the strided kernel)."""
    from test_gpu_strided import direct_column
    rng = np.random.default_rng(27)
    d_stream, d_vals, d_entries, _ = direct_column(rng, 45, 17, True)
    stream, vals, starts = delta_column(rng, 45, 10, True)
    check(ctx, oracle, d_stream, d_vals.size, True, d_entries, PATH_STRIDED, d_vals)
    for _ in range(2):
        check(ctx, oracle, stream, vals.size, True, every(starts, 20), PATH_PLANNED, vals)
        check(ctx, oracle, stream, vals.size, True, every(starts, 20), PATH_PLANNED, vals)
        check(ctx, oracle, d_stream, d_vals.size, True, d_entries, None, d_vals)
        assert ctx.last_path() in (PATH_STRIDED, PATH_PLANNED)
        check(ctx, oracle, d_stream, d_vals.size, True, d_entries, PATH_STRIDED, d_vals)


def test_unaligned_device_output(ctx, oracle):
    """Device-pointer call with an output that is only 8-byte aligned: still the planned path."""
    import torch
    rng = np.random.default_rng(28)
    stream, vals, starts = delta_column(rng, 64, 10, True)
    entries = every(starts, 16)
    d_stream = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy()).cuda()
    d_entries = torch.from_numpy(entries.view(np.int64).copy()).cuda()
    buf = torch.zeros(vals.size + 2, dtype=torch.int64, device="cuda")
    for shift in (0, 1):
        out = buf[shift:shift + vals.size]
        out.zero_()
        ctx.rlev2_device(d_stream.data_ptr(), d_stream.numel(), vals.size, True, out.data_ptr(),
                         d_entries.data_ptr(), d_entries.numel())
        ctx.synchronize()
        assert (out.cpu().numpy() == vals).all()
        assert ctx.last_path() == PATH_PLANNED


@pytest.mark.parametrize("n_runs,cut", [(400, 0), (401, 0), (417, 100), (1000, 0)])
def test_planned_then_broken(ctx, oracle, n_runs, cut):
    """Longer columns, decoded repeatedly on one context; then the same column with one run near
    its end claiming 511 values: the plan fails and the general path gives the oracle's answer."""
    rng = np.random.default_rng(n_runs + cut)
    stream, vals, starts = delta_column(rng, n_runs, 10, True, small_steps=0.2)
    for _ in range(3):
        check(ctx, oracle, stream, vals.size - cut, True, every(starts, 5), PATH_PLANNED, vals)
    broken = bytearray(stream)
    broken[starts[n_runs - 30] + 1] = 0xFE   # that run now claims 511 values
    out = None
    try:
        out = ctx.rlev2(bytes(broken), vals.size - cut, True, every(starts, 5))
    except Exception:
        pass
    rc, ref, _ = oracle.rlev2(bytes(broken), vals.size - cut, True)
    if rc == 0:
        assert out is not None and (out == ref).all()
    else:
        assert out is None
    assert ctx.last_path() == PATH_GENERAL
