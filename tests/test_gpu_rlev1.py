"""Integer RLE version 1 (files written as ORC 0.11) on the GPU against the oracle, whose values
the ORC v1 specification's examples and pyarrow-read files pin (tests/test_oracle.py).  The
reference has no RLE v1 (README.md:23-25); parity with it is not claimed, the contract is that of
its RLE v2 iterators (rle_v2.rs:517-562): num_values values, trailing bytes ignored."""
import numpy as np
import pytest

import rle_encode as enc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from orc_format_amd import default_context
    return default_context()


def decode(ctx, stream, n, signed, entries=None):
    from orc_format_amd.error import OrcError
    try:
        return 0, ctx.rlev1(stream, n, signed, entries)
    except OrcError as e:
        return e.code, None


def test_specification_examples(ctx):
    assert ctx.rlev1(bytes([0x61, 0x00, 0x07]), 100, False).tolist() == [7] * 100
    assert ctx.rlev1(bytes([0x61, 0xFF, 0x64]), 100, False).tolist() == list(range(100, 0, -1))
    assert ctx.rlev1(bytes([0xFB, 0x02, 0x03, 0x06, 0x07, 0x0B]), 5, False).tolist() == [2, 3, 6, 7, 11]
    assert ctx.rlev1(bytes([0xFE, 0x01, 0x04]), 2, True).tolist() == [-1, 2]
    assert ctx.rlev1(bytes([0x61, 0x00, 0x07]), 10, False).tolist() == [7] * 10      # cut inside a run
    assert decode(ctx, bytes([0x61, 0x00]), 3, False)[0] == 1                        # no base
    assert decode(ctx, bytes([0xFB, 0x02, 0x03]), 5, False)[0] == 1                  # literals cut short
    assert decode(ctx, bytes([0x61, 0x00, 0x07]), 101, False)[0] == 1                # more values than there are
    assert decode(ctx, bytes([0xFF] + [0x80] * 10 + [0x01]), 1, False)[0] == 1      # an 11-byte varint


@pytest.mark.parametrize("signed", [False, True])
def test_random_columns_with_and_without_entry_points(ctx, oracle, signed):
    rng = np.random.default_rng(50 + signed)
    for case in range(24):
        n = int(rng.integers(1, 40_000))
        kind = case % 4
        if kind == 0:
            v = rng.integers(-2**62 if signed else 0, 2**62, size=n)                  # literals, long varints
        elif kind == 1:
            v = np.cumsum(rng.integers(-3 if signed else 0, 4, size=n)) + (0 if signed else 7)   # short runs, literals
        elif kind == 2:
            v = np.repeat(rng.integers(0, 50, size=n // 200 + 1), 200)[:n]           # runs of 130 and their rests
        else:
            v = np.arange(n) * int(rng.integers(-128 if signed else 0, 128)) + (0 if signed else 128 * n)
        stream, starts = enc.rlev1(v, signed)
        expect = np.asarray(v, dtype=np.int64)
        for entries in (None, np.array(starts[::7], dtype=np.uint64), np.array(starts, dtype=np.uint64)):
            for want in {n, max(1, n // 2), max(1, n - 1)}:
                rc_cpu, ref, _ = oracle.rlev1(stream, want, signed)
                rc, out = decode(ctx, stream, want, signed, entries)
                assert rc_cpu == 0 and rc == 0, (case, want)
                assert (out == ref).all() and (out == expect[:want]).all(), (case, want)
        # more values than the stream holds, a cut stream, an entry point that is no group start
        assert decode(ctx, stream, n + 1, signed)[0] == 1
        if len(stream) > 3:
            cut = stream[:len(stream) - 1]
            assert (decode(ctx, cut, n, signed)[0] == 0) == (oracle.rlev1(cut, n, signed)[0] == 0)
        if len(starts) > 2 and starts[1] + 1 not in starts:
            bad = np.array([0, starts[1] + 1], dtype=np.uint64)
            rc, out = decode(ctx, stream, n, signed, bad)
            assert rc == 1 or (out == expect).all()


def test_mutated_streams_agree_with_the_oracle(ctx, oracle):
    from test_gpu_fuzz import _mutations
    rng = np.random.default_rng(60)
    ok = bad = 0
    for case in range(60):
        signed = bool(case & 1)
        v = np.cumsum(rng.integers(-2, 3, size=int(rng.integers(1, 600)))) if case % 3 else \
            rng.integers(0, 2**40, size=int(rng.integers(1, 600)))
        if not signed:
            v = np.abs(v)
        stream, _ = enc.rlev1(v, signed)
        for m in _mutations(rng, stream, 6):
            for n in {len(v), len(v) // 2 + 1, len(v) + 9}:
                rc_cpu, ref, _ = oracle.rlev1(m, n, signed)
                rc, out = decode(ctx, m, n, signed)
                assert (rc_cpu == 0) == (rc == 0), (case, n, m.hex()[:120])
                if rc == 0:
                    assert (out == ref).all()
                    ok += 1
                else:
                    bad += 1
    assert ok > 100 and bad > 50


def test_columns_of_a_file_written_as_version_0_11(tmp_path):
    """Integer columns of a file the ORC C++ writer wrote with file version 0.11 (ColumnEncoding
    DIRECT, RLE v1, ROW_INDEX positions as entry points), with nulls: what pyarrow reads."""
    pa = pytest.importorskip("pyarrow")
    orc = pytest.importorskip("pyarrow.orc")
    from orc_format_amd import Context, read
    rng = np.random.default_rng(7)
    rows = 150_000
    table = pa.table({
        "a": pa.array(np.cumsum(rng.integers(-5, 9, size=rows))),
        "b": pa.array(rng.integers(-2**45, 2**45, size=rows), mask=rng.random(rows) < 0.2),
        "c": pa.array(np.repeat(rng.integers(0, 9, size=rows // 10), 10).astype(np.int32)),
        "d": pa.array(rng.standard_normal(rows)),
    })
    path = str(tmp_path / "v011.orc")
    orc.write_table(table, path, compression="uncompressed", file_version="0.11", stripe_size=1 << 20)
    want = orc.ORCFile(path).read()
    md = read.read_metadata(path)
    assert len(md.stripes) >= 2
    first = 0
    with Context(0) as ctx:
        for si in range(len(md.stripes)):
            footer = read.read_stripe_footer(md, si)
            cols = ctx.decode_stripe(md, si, footer, [1, 2, 3, 4])
            n = cols[0]["validity"].size
            part = want.slice(first, n)
            for ci, name in enumerate(["a", "b", "c"]):
                ref = part.column(name)
                assert cols[ci]["validity"].tolist() == ref.is_valid().to_pylist()
                assert (cols[ci]["values"] == np.array(ref.drop_null().to_pylist(), dtype=np.int64)).all(), (si, name)
            assert np.array_equal(cols[3]["values"], np.array(part.column("d").to_pylist()))
            first += n
    assert first == rows


def test_string_columns_of_a_file_written_as_version_0_11(tmp_path):
    """Direct and dictionary string columns of a version-0.11 file: LENGTH and the dictionary
    indices are RLE v1 there; with nulls, empty strings and multi-byte characters."""
    pa = pytest.importorskip("pyarrow")
    orc = pytest.importorskip("pyarrow.orc")
    from orc_format_amd import Context, read
    rng = np.random.default_rng(8)
    rows = 60_000
    words = ["", "a", "bb", "ccc", "häuser", "日本語", "x" * 40] + ["w%d" % k for k in range(200)]
    table = pa.table({
        "direct": pa.array(["%x-%d" % (k, k % 7) for k in rng.integers(0, 2**50, size=rows)],
                           mask=rng.random(rows) < 0.1),
        "dict": pa.array([words[k] for k in rng.integers(0, len(words), size=rows)], mask=rng.random(rows) < 0.2),
    })
    for threshold in (0.0, 1.0):   # everything direct / dictionaries where they pay
        path = str(tmp_path / ("v011s_%d.orc" % int(threshold)))
        orc.write_table(table, path, compression="uncompressed", file_version="0.11", stripe_size=1 << 20,
                        dictionary_key_size_threshold=threshold)
        want = orc.ORCFile(path).read()
        md = read.read_metadata(path)
        first = 0
        encodings = set()
        with Context(0) as ctx:
            for si in range(len(md.stripes)):
                footer = read.read_stripe_footer(md, si)
                for col in (1, 2):
                    encodings.add(int(read.read_stripe_column(md, si, footer, col).encoding()))
                cols = ctx.decode_stripe(md, si, footer, [1, 2])
                n = cols[0]["validity"].size
                part = want.slice(first, n)
                for ci, name in enumerate(["direct", "dict"]):
                    ref = part.column(name)
                    assert cols[ci]["validity"].tolist() == ref.is_valid().to_pylist()
                    raw, off = cols[ci]["bytes"].tobytes(), cols[ci]["offsets"]
                    got = [raw[off[i]:off[i + 1]].decode() for i in range(len(off) - 1)]
                    assert got == ref.drop_null().to_pylist(), (threshold, si, name)
                first += n
        assert first == rows
        assert encodings <= {0, 1}, encodings   # DIRECT / DICTIONARY: the version-1 encodings
    assert True
