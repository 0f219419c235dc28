"""Pins the CPU oracle: the reference's own known-answer vectors, the decoded columns its
integration tests expect, and (beyond what the reference decodes) the Apache ORC C++ reader."""
import json
import os

import numpy as np
import pytest

import assemble
import orc_pyparse as op


def hx(s):
    return bytes.fromhex(s.replace(" ", ""))


# ---- unit vectors of src/read/decode/rle_v2.rs:764-815 ----------------------------------------

def test_zigzag(oracle):
    assert oracle.zigzag(2) == 1
    assert oracle.zigzag(4) == 2
    assert oracle.zigzag(1) == -1
    assert oracle.zigzag(2**64 - 1) == -(2**63)


def test_unpacking(oracle):
    assert oracle.unpack(bytes([0b01000000]), 2, 0) == 1
    assert oracle.unpack(bytes([0b01000000]), 2, 1) == 0
    # a width that is not a whole number of bytes, straddling a byte (index 2 of width 3)
    assert oracle.unpack(bytes([0b00000010, 0b10000000]), 3, 2) == 0b101


def test_short_repeat(oracle):
    rc, out, used = oracle.rlev2(hx("0a 27 10"), 5, False)
    assert rc == 0 and used == 3
    assert out.tolist() == [10000] * 5


def test_direct(oracle):
    rc, out, used = oracle.rlev2(hx("5e 03 5c a1 ab 1e de ad be ef"), 4, False)
    assert rc == 0 and used == 10
    assert out.tolist() == [23713, 43806, 57005, 48879]


def test_delta(oracle):
    rc, out, used = oracle.rlev2(hx("c6 09 02 02 22 42 42 46"), 10, False)
    assert rc == 0 and used == 8
    assert out.tolist() == [2, 3, 5, 7, 11, 13, 17, 19, 23, 29]


def test_patched_base_spec_example(oracle):
    # ORC v1 specification, "Patched Base" example
    data = hx("8e 13 2b 21 07 d0 1e 00 14 70 28 32 3c 46 50 5a 64 6e 78 82 8c 96 a0 aa b4 be fc e8")
    expect = [2030, 2000, 2020, 1000000, 2040, 2050, 2060, 2070, 2080, 2090, 2100, 2110, 2120,
              2130, 2140, 2150, 2160, 2170, 2180, 2190]
    rc, out, used = oracle.rlev2(data, 20, False)
    assert rc == 0 and used == len(data)
    assert out.tolist() == expect


# ---- unit vectors of src/read/decode/boolean_rle.rs:164-208 -----------------------------------

def test_boolean_basic(oracle):
    rc, out, ones = oracle.boolean(hx("61 00"), 100)
    assert rc == 0 and ones == 0 and out.tolist() == [False] * 100


def test_boolean_literals(oracle):
    rc, out, ones = oracle.boolean(bytes([0xFE, 0b01000100, 0b01000101]), 16)
    assert rc == 0
    assert out.tolist() == [bool(int(c)) for c in "0100010001000101"]
    assert ones == 5


def test_boolean_another(oracle):
    rc, out, _ = oracle.boolean(hx("ff 80"), 8)
    assert rc == 0 and out.tolist() == [True] + [False] * 7


def test_byte_rle_run_is_exact(oracle):
    # a run of 3 x 0xff followed by a literal 0x00: 24 ones then zeros
    rc, out, ones = oracle.boolean(hx("00 ff ff 00"), 32)
    assert rc == 0 and ones == 24 and out[:24].all() and not out[24:].any()


def test_truncated_streams(oracle):
    assert oracle.rlev2(hx("5e 03 5c a1"), 4, False)[0] == 1
    assert oracle.rlev2(b"", 1, True)[0] == 1
    assert oracle.boolean(hx("fe 44"), 16)[0] == 1
    assert oracle.floats(b"\0" * 7, 2, np.float32)[0] == 5
    assert oracle.rlev2(b"", 0, True)[0] == 0


# ---- stream bytes written by the ORC C++ writer for the rows of write.py:5-23 -----------------

GOLDEN_STREAMS = [
    # (hex, n, signed, expected)
    ("46 03 16 32", 4, False, [1, 6, 3, 2]),        # c3 LENGTH, Direct w4
    ("c2 03 01 02 40", 4, False, [1, 2, 3, 3]),     # c4 LENGTH, Delta w2 +
    ("c2 03 03 01 10", 4, False, [3, 2, 2, 1]),     # c5 LENGTH, Delta w2 -
    ("01 05", 4, False, [5, 5, 5, 5]),              # c6 LENGTH, ShortRepeat
    ("01 0a", 4, True, [5, 5, 5, 5]),               # c7 DATA
    ("01 09", 4, True, [-5, -5, -5, -5]),           # c8 DATA
    ("c2 03 02 02 90", 4, True, [1, 2, 4, 5]),      # c9 DATA
    ("c2 03 0a 01 90", 4, True, [5, 4, 2, 1]),      # c10 DATA
    ("46 03 2c 64", 4, True, [1, 6, 3, 2]),         # c11/c13 DATA
    ("46 03 1b 53", 4, True, [-1, -6, -3, -2]),     # c12/c14 DATA
    ("46 04 a9 2a a0", 5, True, [5, -5, 1, 5, 5]),  # c15 DATA
    ("c0 04 01 02", 5, False, [1, 2, 3, 4, 5]),     # c16 LENGTH, Delta w0 fixed +1
    ("c0 04 05 01", 5, False, [5, 4, 3, 2, 1]),     # c17 LENGTH
    ("c0 3f 04 00", 64, False, [4] * 64),           # string_long LENGTH
]


@pytest.mark.parametrize("hexs,n,signed,expect", GOLDEN_STREAMS)
def test_golden_streams(oracle, hexs, n, signed, expect):
    rc, out, used = oracle.rlev2(hx(hexs), n, signed)
    assert rc == 0 and used == len(hx(hexs))
    assert out.tolist() == expect


def test_golden_present(oracle):
    rc, out, ones = oracle.boolean(hx("ff d8"), 5)
    assert rc == 0 and out.tolist() == [True, True, False, True, True] and ones == 4
    rc, out, _ = oracle.boolean(hx("ff a0"), 4)
    assert out.tolist() == [True, False, True, False]


# ---- the decoded columns tests/it/main.rs expects ----------------------------------------------

@pytest.fixture(scope="module")
def expected(golden_dir):
    with open(os.path.join(golden_dir, "expected.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("col", range(1, 18))
def test_fixture_test_orc(oracle, golden_dir, expected, col):
    f = op.OrcFile(os.path.join(golden_dir, "test.orc"))
    valid, vals = assemble.column(oracle, f, 0, col)
    exp = expected["test.orc"]
    assert valid.tolist() == (exp["validity_1_14"] if col <= 14 else exp["validity_15_17"])
    got = vals.tolist() if isinstance(vals, np.ndarray) else vals
    assert got == exp[str(col)]


def test_fixture_run_kinds(oracle, golden_dir):
    """The sub-encodings the reference's test names promise (main.rs:88-176)."""
    f = op.OrcFile(os.path.join(golden_dir, "test.orc"))
    kinds = {7: 0, 8: 0, 9: 3, 10: 3, 11: 1, 12: 1, 13: 1, 14: 1, 15: 1}
    for col, kind in kinds.items():
        n = 5 if col == 15 else 4
        rc, runs = oracle.runs(f.stream(0, col, op.DATA), n)
        assert rc == 0 and [r[0] for r in runs] == [kind], (col, runs)


@pytest.mark.parametrize("name", ["long_bool.orc", "string_long.orc", "string_dict.orc",
                                  "string_long_long.orc"])
def test_fixture_long_files(oracle, golden_dir, expected, name):
    f = op.OrcFile(os.path.join(golden_dir, name))
    exp = expected[name]
    valid, vals = assemble.column(oracle, f, 0, 1)
    assert valid.tolist() == [True] * exp["rows"]
    if "1" in exp:
        assert vals.tolist() == exp["1"]
    else:
        cyc = exp["1_cycle"]
        assert vals == [cyc[i % len(cyc)] for i in range(exp["rows"])]


def test_string_dict_is_dictionary_encoded(golden_dir):
    f = op.OrcFile(os.path.join(golden_dir, "string_dict.orc"))
    _, enc = f.stripe_footer(0)
    assert enc[1] == (assemble.DICTIONARY_V2, 2)


# ---- beyond the reference: Apache ORC C++ reader as arbiter -----------------------------------

@pytest.mark.parametrize("name", ["mixed_compression.orc", "mixed_speed.orc"])
def test_mixed_against_pyarrow(oracle, golden_dir, name):
    orc = pytest.importorskip("pyarrow.orc")
    path = os.path.join(golden_dir, name)
    table = orc.ORCFile(path).read()
    f = op.OrcFile(path)
    assert len(f.stripes) > 1 or name == "mixed_compression.orc"  # one big stripe / many small
    kinds_seen = set()
    for ci, cname in enumerate(table.column_names):
        col = ci + 1
        got = []
        for s in range(len(f.stripes)):
            valid, vals = assemble.column(oracle, f, s, col)
            got += assemble.scatter(valid, vals)
            if f.types[col][0] in (assemble.INT, assemble.LONG):
                rc, runs = oracle.runs(f.stream(s, col, op.DATA), int(valid.sum()))
                kinds_seen |= {(r[0], r[1]) for r in runs}
        exp = table.column(cname).to_pylist()
        if f.types[col][0] == assemble.DOUBLE:
            a = np.array(got, dtype=np.float64).view(np.uint64)
            b = np.array(exp, dtype=np.float64).view(np.uint64)
            assert (a == b).all(), cname
        else:
            assert got == exp, cname
    run_kinds = {k for k, _ in kinds_seen}
    assert run_kinds == {0, 1, 2, 3}, run_kinds  # ShortRepeat, Direct, PatchedBase, Delta all met
    if name == "mixed_compression.orc":
        assert (1, 17) in kinds_seen  # Direct with a 17-bit width


# ---- the test-side stream writer agrees with the oracle (all kinds, all widths) -----------------

@pytest.mark.parametrize("signed", [False, True])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_streams_roundtrip(oracle, signed, seed):
    import rle_encode as enc
    rng = np.random.default_rng(seed)
    stream, expect, starts = enc.random_stream(rng, 200, signed)
    rc, out, used = oracle.rlev2(stream, expect.size, signed)
    assert rc == 0 and used == len(stream)
    assert (out == expect).all()
    rc, runs = oracle.runs(stream, expect.size)
    assert [r[3] for r in runs] == starts


def test_byte_rle_writer_roundtrip(oracle):
    import rle_encode as enc
    rng = np.random.default_rng(5)
    data = bytearray(rng.integers(0, 256, size=5000, dtype=np.uint8).tobytes())
    data[100:400] = b"\xff" * 300
    data[1000:1003] = b"\x00" * 3
    stream = enc.byte_rle(data)
    rc, out = oracle.byte_rle(stream, len(data))
    assert rc == 0 and out.tobytes() == bytes(data)


def test_rlev1_specification_examples(oracle):
    """Integer RLE version 1, the examples of the ORC v1 specification (the reference has no
    RLE v1: README.md:23-25): a hundred sevens, 100 down to 1, and five literals."""
    rc, out, used = oracle.rlev1(bytes([0x61, 0x00, 0x07]), 100, False)
    assert rc == 0 and used == 3 and out.tolist() == [7] * 100
    rc, out, used = oracle.rlev1(bytes([0x61, 0xFF, 0x64]), 100, False)
    assert rc == 0 and out.tolist() == list(range(100, 0, -1))
    rc, out, used = oracle.rlev1(bytes([0xFB, 0x02, 0x03, 0x06, 0x07, 0x0B]), 5, False)
    assert rc == 0 and used == 6 and out.tolist() == [2, 3, 6, 7, 11]
    # signed streams zigzag their varints; fewer values than the group holds; a cut stream
    rc, out, _ = oracle.rlev1(bytes([0xFE, 0x01, 0x04]), 2, True)
    assert rc == 0 and out.tolist() == [-1, 2]
    rc, out, _ = oracle.rlev1(bytes([0x61, 0x00, 0x07]), 10, False)
    assert rc == 0 and out.tolist() == [7] * 10
    assert oracle.rlev1(bytes([0x61, 0x00]), 3, False)[0] == 1
    assert oracle.rlev1(bytes([0xFB, 0x02, 0x03]), 5, False)[0] == 1
    assert oracle.rlev1(bytes([0x61, 0x00, 0x07]), 101, False)[0] == 1


def test_rlev1_encoder_round_trip(oracle):
    import rle_encode as enc
    rng = np.random.default_rng(4)
    for signed in (False, True):
        for case in range(30):
            n = int(rng.integers(1, 3000))
            kind = case % 3
            if kind == 0:
                v = rng.integers(-2**40 if signed else 0, 2**40, size=n)
            elif kind == 1:
                v = np.cumsum(rng.integers(-3 if signed else 0, 4, size=n)) + (0 if signed else 10)
                v = np.abs(v) if not signed else v
            else:
                v = np.repeat(rng.integers(0, 50, size=n // 7 + 1), 7)[:n]
            stream, starts = enc.rlev1(v, signed)
            rc, out, used = oracle.rlev1(stream, len(v), signed)
            assert rc == 0 and used == len(stream) and (out == np.asarray(v, dtype=np.int64)).all()


def test_rlev1_against_files_written_as_version_0_11(tmp_path, oracle):
    """Columns of a file the ORC C++ writer wrote with file version 0.11 (ColumnEncoding DIRECT,
    RLE v1): the oracle's values are what pyarrow reads."""
    pa = pytest.importorskip("pyarrow")
    orc = pytest.importorskip("pyarrow.orc")
    import orc_pyparse as op
    rng = np.random.default_rng(6)
    rows = 20_000
    cols = {"a": np.cumsum(rng.integers(-5, 9, size=rows)), "b": rng.integers(-2**45, 2**45, size=rows),
            "c": np.repeat(rng.integers(0, 9, size=rows // 10), 10)}
    path = str(tmp_path / "v011.orc")
    orc.write_table(pa.table({k: pa.array(v) for k, v in cols.items()}), path, compression="uncompressed",
                    file_version="0.11")
    f = op.OrcFile(path)
    for ci, name in enumerate(cols, start=1):
        got = []
        for si in range(len(f.stripes)):
            data = f.stream(si, ci, op.DATA)
            n = f.stripes[si].number_of_rows
            rc, out, _ = oracle.rlev1(data, n, True)
            assert rc == 0
            got.append(out)
        assert (np.concatenate(got) == cols[name]).all(), name
