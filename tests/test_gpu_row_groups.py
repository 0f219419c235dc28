"""Entry points that are the ROW_INDEX positions of a file: one per row group of S values, S no
multiple of 512, so that a row group starts somewhere inside a run and the segments between entry
points hold unequal numbers of runs (src/proto.rs:89-99; the writers record a position every
10 000 rows without cutting the run).  With the number of values per group
(orcd_rlev2_decode_groups) the columns of full runs take the paths that need no header scan; the
answers are the oracle's and pyarrow's either way (rle_v2.rs:135-175, 184-272, 331-415, 480-514)."""
import os

import numpy as np
import pytest

import rle_encode as enc

pytestmark = pytest.mark.gpu

PATH_GENERAL, PATH_STRIDED, PATH_PLANNED = 0, 1, 2


@pytest.fixture()
def ctx():
    from orc_format_amd import Context
    with Context(0) as c:
        yield c


def group_entries(starts, n_values, stride):
    """Entry k = header of the run that holds value k * stride (runs of 512 values)."""
    ks = range((n_values + stride - 1) // stride)
    return np.array([starts[k * stride // 512] for k in ks], dtype=np.uint64)


def direct_runs(rng, n_runs, width, signed, last_len=512):
    parts, starts, vals, pos = [], [], [], 0
    for r in range(n_runs):
        n = last_len if r == n_runs - 1 else 512
        hi = 1 << (width - 1 if signed else width)
        v = rng.integers(-hi if signed else 0, hi, size=n, dtype=np.int64)
        v[0] = hi - 1  # the width is exactly this
        b = enc.direct(v if signed else v.astype(np.uint64), width, signed)
        starts.append(pos)
        pos += len(b)
        parts.append(b)
        vals.append(v)
    return b"".join(parts), np.concatenate(vals), starts


def check(ctx, oracle, stream, n, signed, entries, stride, want_path, expect=None):
    out = ctx.rlev2(stream, n, signed, entries, values_per_group=stride)
    rc, ref, _ = oracle.rlev2(stream, n, signed)
    assert rc == 0
    assert (out == ref).all()
    if expect is not None:
        assert (out == expect[:n]).all()
    if want_path is not None:
        assert ctx.last_path() == want_path


@pytest.mark.parametrize("stride", [10000, 1000, 512, 777, 5120, 65536])
@pytest.mark.parametrize("width,signed", [(17, True), (8, False), (32, True)])
def test_direct_column_with_row_groups(ctx, oracle, stride, width, signed):
    rng = np.random.default_rng(stride + width)
    n_runs = max(45, 3 * stride // 512 + 5)
    stream, vals, starts = direct_runs(rng, n_runs, width, signed, last_len=300)
    entries = group_entries(starts, vals.size, stride)
    check(ctx, oracle, stream, vals.size, signed, entries, stride, PATH_STRIDED, vals)
    # the values wanted may end inside the last run, or inside an earlier one
    check(ctx, oracle, stream, vals.size - 123, signed, entries, stride, PATH_STRIDED, vals)


@pytest.mark.parametrize("stride", [10000, 1000, 777, 5120])
@pytest.mark.parametrize("width", [5, 10, 16, 24, 40])
def test_delta_column_with_row_groups(ctx, oracle, stride, width):
    from test_gpu_planned import delta_column
    rng = np.random.default_rng(stride * 3 + width)
    n_runs = max(61, 3 * stride // 512 + 7)
    stream, vals, starts = delta_column(rng, n_runs, width, True, small_steps=0.3, last_len=77)
    entries = group_entries(starts, vals.size, stride)
    check(ctx, oracle, stream, vals.size, True, entries, stride, PATH_PLANNED, vals)
    check(ctx, oracle, stream, vals.size, True, entries, stride, PATH_PLANNED, vals)


def test_descending_and_unsigned_delta(ctx, oracle):
    from test_gpu_planned import delta_column
    rng = np.random.default_rng(5)
    stream, vals, starts = delta_column(rng, 90, 12, False, descending=True, small_steps=0.5)
    entries = group_entries(starts, vals.size, 10000)
    check(ctx, oracle, stream, vals.size, False, entries, 10000, PATH_PLANNED, vals)


def test_wrong_group_size_is_only_slower(ctx, oracle):
    """A number of values per group that the entry points do not bear out: same values, general path."""
    rng = np.random.default_rng(6)
    stream, vals, starts = direct_runs(rng, 120, 17, True)
    entries = group_entries(starts, vals.size, 10000)
    for wrong in (9000, 10240, 20000, 513):
        check(ctx, oracle, stream, vals.size, True, entries, wrong, PATH_GENERAL, vals)


def test_a_short_run_in_a_group_is_only_slower(ctx, oracle):
    """One run of 400 values: every later row group starts in another run than the arithmetic says.
    (The index of such a column would not carry these skips; the decoder must not rely on that.)"""
    from test_gpu_planned import delta_column
    rng = np.random.default_rng(7)
    a, va, sa = delta_column(rng, 30, 10, True)
    b, vb, sb = delta_column(rng, 1, 10, True, last_len=400)
    c, vc, sc = delta_column(rng, 40, 10, True)
    stream = a + b + c
    vals = np.concatenate([va, vb, vc])
    starts = sa + [len(a)] + [len(a) + len(b) + s for s in sc]
    # entry points at true run headers, as a walk of the chain finds them
    run_first = np.concatenate([[0], np.cumsum([512] * 30 + [400] + [512] * 40)])[:-1]
    ks = range((vals.size + 9999) // 10000)
    entries = np.array([starts[int(np.searchsorted(run_first, k * 10000, side="right")) - 1] for k in ks], dtype=np.uint64)
    check(ctx, oracle, stream, vals.size, True, entries, 10000, PATH_GENERAL, vals)


def test_entry_point_off_its_run_is_an_error(ctx):
    from orc_format_amd.error import OrcError
    rng = np.random.default_rng(8)
    stream, vals, starts = direct_runs(rng, 100, 17, True)
    entries = group_entries(starts, vals.size, 10000)
    entries[2] += 1
    with pytest.raises(OrcError) as e:
        ctx.rlev2(stream, vals.size, True, entries, values_per_group=10000)
    assert e.value.code == 1


def test_mixed_widths_between_groups(ctx, oracle):
    """Direct runs whose width changes from one row group to the next: not for the strided kernel
    (run sizes differ), but every run holds 512 values and the planned path finds them."""
    rng = np.random.default_rng(9)
    parts, starts, vals, pos = [], [], [], 0
    for r in range(118):
        width = (9, 17, 12)[(r * 512 // 10000) % 3]
        v = rng.integers(0, 1 << (width - 1), size=512, dtype=np.int64)
        v[0] = (1 << (width - 1)) - 1
        b = enc.direct(v, width, True)
        starts.append(pos)
        pos += len(b)
        parts.append(b)
        vals.append(v)
    stream, vals = b"".join(parts), np.concatenate(vals)
    entries = group_entries(starts, vals.size, 10000)
    out = ctx.rlev2(stream, vals.size, True, entries, values_per_group=10000)
    assert (out == vals).all()
    rc, ref, _ = oracle.rlev2(stream, vals.size, True)
    assert rc == 0 and (out == ref).all()


def test_file_written_by_the_orc_writer(ctx, tmp_path):
    """A file of the Apache ORC C++ writer (pyarrow): 17-bit Direct runs and Delta runs of sorted
    timestamps, row index stride 10 000.  The index's values-to-skip give the group size, the
    columns decode without a header scan, and the values are what pyarrow's reader returns."""
    pa = pytest.importorskip("pyarrow")
    orc = pytest.importorskip("pyarrow.orc")
    from orc_format_amd import read
    rng = np.random.default_rng(10)
    n = 6_000_000
    table = pa.table({"a": pa.array(rng.integers(0, 2**16, size=n)),
                      "ts": pa.array(1_600_000_000_000 + np.cumsum(rng.integers(1, 1000, size=n))),
                      "mixed": pa.array(np.where(rng.random(n) < 0.001, 2**40, rng.integers(0, 100, size=n)))})
    path = os.path.join(tmp_path, "written.orc")
    orc.write_table(table, path, compression="uncompressed", compression_strategy="compression",
                    stripe_size=8 << 20)
    f = orc.ORCFile(path)
    md = read.read_metadata(path)
    assert len(md.stripes) == f.nstripes
    for stripe in range(f.nstripes):
        want = f.read_stripe(stripe)
        footer = read.read_stripe_footer(md, stripe)
        for ci, name, path_wanted in ((1, "a", (PATH_STRIDED, PATH_PLANNED)), (2, "ts", (PATH_PLANNED,)), (3, "mixed", None)):
            col = read.read_stripe_column(md, stripe, footer, ci)
            if path_wanted:
                assert col.values_per_group(1) == 10000
            got = ctx.decode_column(md, stripe, footer, ci)
            assert np.array_equal(got["values"], want.column(name).to_numpy())
            if path_wanted:
                assert ctx.last_path() in path_wanted, (stripe, name, ctx.last_path())
        # the three columns in one call: the long regular ones by themselves, the rest batched
        cols = ctx.decode_stripe(md, stripe, footer, [1, 2, 3])
        for got, name in zip(cols, ("a", "ts", "mixed")):
            assert np.array_equal(got["values"], want.column(name).to_numpy())
