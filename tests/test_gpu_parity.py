"""Parity of the HIP decoder (called through the C ABI) with the CPU oracle: the reference's
known-answer vectors, the fixture files, random streams of every run kind and bit width, error
behaviour, device-pointer calls and size-independent properties at larger sizes."""
import json
import os

import numpy as np
import pytest

import assemble
import orc_pyparse as op
import rle_encode as enc
from test_oracle import GOLDEN_STREAMS, hx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from gpu_backend import GpuBackend
    return GpuBackend()


@pytest.fixture(scope="module")
def expected(golden_dir):
    with open(os.path.join(golden_dir, "expected.json")) as f:
        return json.load(f)


# ---- known-answer vectors (rle_v2.rs:764-815, boolean_rle.rs:164-208) --------------------------

def test_kat_rlev2(gpu):
    assert gpu.rlev2(hx("0a 27 10"), 5, False)[1].tolist() == [10000] * 5
    assert gpu.rlev2(hx("5e 03 5c a1 ab 1e de ad be ef"), 4, False)[1].tolist() == \
        [23713, 43806, 57005, 48879]
    assert gpu.rlev2(hx("c6 09 02 02 22 42 42 46"), 10, False)[1].tolist() == \
        [2, 3, 5, 7, 11, 13, 17, 19, 23, 29]
    data = hx("8e 13 2b 21 07 d0 1e 00 14 70 28 32 3c 46 50 5a 64 6e 78 82 8c 96 a0 aa b4 be fc e8")
    assert gpu.rlev2(data, 20, False)[1].tolist() == \
        [2030, 2000, 2020, 1000000] + list(range(2040, 2200, 10))


def test_kat_boolean(gpu):
    rc, out, ones = gpu.boolean(hx("61 00"), 100)
    assert rc == 0 and ones == 0 and out.tolist() == [False] * 100
    rc, out, ones = gpu.boolean(bytes([0xFE, 0b01000100, 0b01000101]), 16)
    assert out.tolist() == [bool(int(c)) for c in "0100010001000101"] and ones == 5
    rc, out, _ = gpu.boolean(hx("ff 80"), 8)
    assert out.tolist() == [True] + [False] * 7
    rc, out, ones = gpu.boolean(hx("00 ff ff 00"), 32)
    assert ones == 24 and out[:24].all() and not out[24:].any()


@pytest.mark.parametrize("hexs,n,signed,expect", GOLDEN_STREAMS)
def test_golden_streams(gpu, hexs, n, signed, expect):
    rc, out, _ = gpu.rlev2(hx(hexs), n, signed)
    assert rc == 0 and out.tolist() == expect


# ---- fixture files: tests/it/main.rs expectations and the oracle ------------------------------

@pytest.mark.parametrize("col", range(1, 18))
def test_fixture_test_orc(gpu, golden_dir, expected, col):
    f = op.OrcFile(os.path.join(golden_dir, "test.orc"))
    valid, vals = assemble.column(gpu, f, 0, col)
    exp = expected["test.orc"]
    assert valid.tolist() == (exp["validity_1_14"] if col <= 14 else exp["validity_15_17"])
    got = vals.tolist() if isinstance(vals, np.ndarray) else vals
    assert got == exp[str(col)]


@pytest.mark.parametrize("name", ["long_bool.orc", "string_long.orc", "string_dict.orc",
                                  "string_long_long.orc"])
def test_fixture_long_files(gpu, golden_dir, expected, name):
    f = op.OrcFile(os.path.join(golden_dir, name))
    exp = expected[name]
    valid, vals = assemble.column(gpu, f, 0, 1)
    assert valid.tolist() == [True] * exp["rows"]
    if "1" in exp:
        assert vals.tolist() == exp["1"]
    else:
        cyc = exp["1_cycle"]
        assert vals == [cyc[i % len(cyc)] for i in range(exp["rows"])]


@pytest.mark.parametrize("name", ["mixed_compression.orc", "mixed_speed.orc"])
def test_mixed_files_match_oracle(gpu, oracle, golden_dir, name):
    f = op.OrcFile(os.path.join(golden_dir, name))
    for s in range(len(f.stripes)):
        for col in range(1, len(f.types)):
            v_gpu, x_gpu = assemble.column(gpu, f, s, col)
            v_cpu, x_cpu = assemble.column(oracle, f, s, col)
            assert (v_gpu == v_cpu).all()
            if isinstance(x_cpu, np.ndarray):
                assert x_gpu.dtype == x_cpu.dtype
                assert (x_gpu.view(np.uint8) == x_cpu.view(np.uint8)).all(), (s, col)
            else:
                assert x_gpu == x_cpu, (s, col)


def test_reference_api_surface(golden_dir, expected):
    """The same call sequence as tests/it/main.rs:8-24 + deserialize.rs:48-82, with the names
    of the reference's API."""
    from orc_format_amd import read
    from orc_format_amd.proto import StreamKind
    from orc_format_amd.read.decode import (BooleanIter, Delta, Direct, ShortRepeat,
                                            SignedRleV2Iter, SignedRleV2RunIter)
    md = read.read_metadata(os.path.join(golden_dir, "test.orc"))
    footer = read.read_stripe_footer(md, 0)
    kinds = {7: ShortRepeat, 9: Delta, 11: Direct}
    for col, klass in kinds.items():
        column = read.read_stripe_column(md, 0, footer, col)
        validity = list(BooleanIter(column.get_stream(StreamKind.Present), column.number_of_rows()))
        assert validity == expected["test.orc"]["validity_1_14"]
        n = sum(validity)
        runs = list(SignedRleV2RunIter(column.get_stream(StreamKind.Data), n))
        assert [type(r) for r in runs] == [klass]
        values = [v for r in runs for v in r]
        assert values == expected["test.orc"][str(col)]
        assert list(SignedRleV2Iter(column.get_stream(StreamKind.Data), n)) == values


def test_entry_points_from_row_index(gpu, oracle, golden_dir):
    """Decoding with the ROW_INDEX entry points gives the same values as without."""
    from orc_format_amd import read
    from orc_format_amd.proto import StreamKind
    path = os.path.join(golden_dir, "mixed_compression.orc")
    md = read.read_metadata(path)
    f = op.OrcFile(path)
    used = 0
    for s in range(len(f.stripes)):
        footer = read.read_stripe_footer(md, s)
        for col in (1, 2, 3, 4, 5):
            column = read.read_stripe_column(md, s, footer, col)
            data = column.get_stream(StreamKind.Data)
            entries = column.entry_points(StreamKind.Data)
            n = column.number_of_rows()
            if column.has_stream(StreamKind.Present):
                present = column.get_stream(StreamKind.Present)
                pe = column.entry_points(StreamKind.Present)
                rc, valid, ones = gpu.boolean(present, n, pe)
                assert rc == 0
                rc, valid_cpu, ones_cpu = oracle.boolean(present, n)
                assert (valid == valid_cpu).all() and ones == ones_cpu
                n = ones
            rc, out, _ = gpu.rlev2(data, n, True, entries)
            assert rc == 0
            rc, ref, _ = oracle.rlev2(data, n, True)
            assert (out == ref).all()
            used += len(entries) > 1
    assert used > 0


# ---- random streams: every kind and width, odd alignments --------------------------------------

@pytest.mark.parametrize("signed", [False, True])
@pytest.mark.parametrize("seed,n_runs,max_len", [(1, 300, 512), (2, 2000, 12), (3, 50, 512),
                                                  (4, 700, 100)])
def test_random_streams(gpu, oracle, signed, seed, n_runs, max_len):
    rng = np.random.default_rng(seed)
    stream, expect, starts = enc.random_stream(rng, n_runs, signed, max_len=max_len)
    rc, ref, _ = oracle.rlev2(stream, expect.size, signed)
    assert rc == 0 and (ref == expect).all()
    rc, out, _ = gpu.rlev2(stream, expect.size, signed)
    assert rc == 0
    bad = np.nonzero(out != expect)[0]
    assert bad.size == 0, (bad[:10], out[bad[:10]], expect[bad[:10]])
    # with entry points: every 7th run start, and all run starts
    for step in (7, 1):
        rc, out, _ = gpu.rlev2(stream, expect.size, signed, np.array(starts[::step], dtype=np.uint64))
        assert rc == 0 and (out == expect).all()
    # the run table agrees with the oracle's
    rc, runs_cpu = oracle.runs(stream, expect.size)
    rc, runs_gpu = gpu.runs(stream, expect.size)
    assert [(k, w, l) for k, w, l, _, _ in runs_gpu] == [(k, w, l) for k, w, l, _, _ in runs_cpu]


@pytest.mark.parametrize("signed", [False, True])
@pytest.mark.parametrize("tail_gap", [1, 64, 65, 300])
def test_irregular_entry_points(gpu, oracle, signed, tail_gap):
    """Many segments of very different sizes: those of up to 64 runs are emitted from the
    descriptors the counting pass kept, longer ones by a second walk, in one run table."""
    rng = np.random.default_rng(500 + tail_gap + signed)
    stream, expect, starts = enc.random_stream(rng, 20000, signed, max_len=40)
    picks, k = [0], 0
    while True:
        k += int(rng.choice([1, 2, 5, 63, 64, 65, 130, 300, 600]))
        if k >= len(starts) - tail_gap:
            break
        picks.append(k)
    picks.append(len(starts) - tail_gap)
    assert len(picks) >= 64
    entries = np.array([starts[i] for i in picks], dtype=np.uint64)
    rc, out, _ = gpu.rlev2(stream, expect.size, signed, entries)
    assert rc == 0 and (out == expect).all()
    # fewer values than the stream holds: the tail segments are not wanted
    for n in (expect.size // 2, expect.size - 1, 513):
        rc, out, _ = gpu.rlev2(stream, n, signed, entries)
        assert rc == 0 and (out == expect[:n]).all()
    # an entry point at the very end of the stream: a last segment without runs
    rc, out, _ = gpu.rlev2(stream, expect.size, signed, np.append(entries, np.uint64(len(stream))))
    assert rc == 0 and (out == expect).all()
    if not signed:
        rc, runs_cpu = oracle.runs(stream, expect.size)
        rc, runs_gpu = gpu.runs(stream, expect.size, entries)
        assert rc == 0 and [(k, w, l) for k, w, l, _, _ in runs_gpu] == [(k, w, l) for k, w, l, _, _ in runs_cpu]


@pytest.mark.parametrize("kind", [1, 3])
@pytest.mark.parametrize("width", enc.WIDTHS)
def test_every_width(gpu, oracle, kind, width):
    if kind == 3 and width == 1:
        pytest.skip("a Delta run cannot carry 1-bit deltas")
    rng = np.random.default_rng(width * 10 + kind)
    # a short odd-length run first, so that the long runs start at odd output offsets
    stream, expect, _ = enc.random_stream(rng, 9, True, kinds=(kind,), widths=[width], min_len=3)
    rc, out, _ = gpu.rlev2(stream, expect.size, True)
    assert rc == 0 and (out == expect).all()
    rc, ref, _ = oracle.rlev2(stream, expect.size, True)
    assert (out == ref).all()


def test_truncate_at_num_values_and_trailing_bytes(gpu, oracle):
    rng = np.random.default_rng(11)
    stream, expect, _ = enc.random_stream(rng, 40, True)
    for n in (1, 2, 511, 512, 513, expect.size - 1):
        rc, out, _ = gpu.rlev2(stream, n, True)
        assert rc == 0 and (out == expect[:n]).all()
    garbage = stream + bytes(rng.integers(0, 256, size=300, dtype=np.uint8))
    rc, out, _ = gpu.rlev2(garbage, expect.size, True)
    assert rc == 0 and (out == expect).all()


def test_errors(gpu, oracle):
    assert gpu.rlev2(hx("5e 03 5c a1"), 4, False)[0] == 1          # truncated Direct run
    assert gpu.rlev2(b"", 1, True)[0] == 1                          # empty stream
    assert gpu.rlev2(b"", 0, True)[0] == 0
    assert gpu.rlev2(hx("0a 27 10"), 6, False)[0] == 1              # stream ends before n values
    assert gpu.rlev2(hx("c6 09 ff ff ff ff ff ff ff ff ff ff ff 02"), 10, False)[0] == 1  # varint
    assert gpu.boolean(hx("fe 44"), 16)[0] == 1
    assert gpu.boolean(b"", 1)[0] == 1
    assert gpu.floats(b"\0" * 7, 2, np.float32)[0] == 5             # DecodeFloat
    # an entry point that is not a run boundary
    rng = np.random.default_rng(12)
    stream, expect, starts = enc.random_stream(rng, 10, True, kinds=(1,), widths=[17], min_len=100)
    bad = np.array([0, starts[3] + 1], dtype=np.uint64)
    assert gpu.rlev2(stream, expect.size, True, bad)[0] == 1
    # LENGTH stream asking for more bytes than DATA holds
    assert gpu.string_direct(hx("01 05"), b"abc", 4)[0] == 1
    # ill-formed UTF-8 and a value boundary inside a code point
    assert gpu.string_direct(hx("01 01"), b"\xff\xfe\xfd\xfc", 4)[0] == 2
    assert gpu.string_direct(hx("46 01 13"), "aé".encode() + b"z", 2, True)[0] == 0
    assert gpu.string_direct(hx("46 01 22"), "aé".encode() + b"z", 2, True)[0] == 2
    assert oracle.string_direct(hx("46 01 22"), "aé".encode() + b"z", 2, True)[0] == 2
    # dictionary index out of range
    assert gpu.string_dictionary([0, 1, 2], b"ab", [0, 1, 2])[0] == 1


# ---- byte-RLE / boolean, floats, strings ---------------------------------------------------------

@pytest.mark.parametrize("seed", [1, 2])
def test_boolean_random(gpu, oracle, seed):
    rng = np.random.default_rng(seed)
    n = 100_003
    bits = rng.random(n) < 0.9
    bits[5000:30000] = True
    bits[40000:40100] = False
    packed = np.packbits(bits)
    stream = enc.byte_rle(packed.tobytes())
    rc, ref, ones_ref = oracle.boolean(stream, n)
    assert rc == 0 and (ref == bits).all()
    rc, out, ones = gpu.boolean(stream, n)
    assert rc == 0 and (out == bits).all() and ones == int(bits.sum())
    from orc_format_amd import default_context
    out2, ones2, bitmap = default_context().boolean(stream, n, want_bitmap=True)
    assert (np.unpackbits(bitmap, bitorder="little")[:n].astype(bool) == bits).all()
    rc, raw = gpu.byte_rle(stream, packed.size)
    assert rc == 0 and (raw == packed).all()


def _byte_rle_run_starts(stream):
    starts, pos = [], 0
    while pos < len(stream):
        starts.append(pos)
        c = stream[pos]
        pos += 2 if c < 128 else 1 + (256 - c)
    return starts


@pytest.mark.parametrize("tail_gap", [1, 64, 65, 300])
def test_byte_rle_irregular_entry_points(gpu, oracle, tail_gap):
    """Byte-RLE / boolean streams cut into many segments of very different sizes: those of up
    to 64 runs are emitted from the words the counting pass kept, longer ones by a second walk."""
    rng = np.random.default_rng(600 + tail_gap)
    # short runs and short literals, so that there are thousands of them
    pieces = []
    for _ in range(20000):
        if rng.random() < 0.5:
            pieces.append(bytes([int(rng.integers(0, 256))]) * int(rng.integers(3, 20)))
        else:
            pieces.append(bytes(rng.integers(0, 256, size=int(rng.integers(1, 9)), dtype=np.uint8)))
    data = b"".join(pieces)
    stream = enc.byte_rle(data, max_literal=int(rng.integers(4, 20)))
    starts = _byte_rle_run_starts(stream)
    assert len(starts) > 3000
    picks, k = [0], 0
    while True:
        k += int(rng.choice([1, 2, 5, 63, 64, 65, 130, 300, 600]))
        if k >= len(starts) - tail_gap:
            break
        picks.append(k)
    picks.append(len(starts) - tail_gap)
    assert len(picks) >= 64
    entries = np.array([starts[i] for i in picks], dtype=np.uint64)
    want = np.frombuffer(data, dtype=np.uint8)
    rc, raw = gpu.byte_rle(stream, want.size, entries)
    assert rc == 0 and (raw == want).all()
    bits = np.unpackbits(want).astype(bool)
    for n in (bits.size, bits.size - 3, bits.size // 2 + 1, 1025):
        rc, out, ones = gpu.boolean(stream, n, entries)
        assert rc == 0 and (out == bits[:n]).all() and ones == int(bits[:n].sum())
    rc, out, ones = gpu.boolean(stream, bits.size, np.append(entries, np.uint64(len(stream))))
    assert rc == 0 and (out == bits).all()
    rc, ref, _ = oracle.boolean(stream, bits.size)
    assert rc == 0 and (ref == bits).all()


def _encode_pieces(pieces):
    """pieces: ('r', byte, n) with 3 <= n <= 130 or ('l', bytes) with 1 <= len <= 128 -> stream, decoded."""
    out, dec = bytearray(), bytearray()
    for p in pieces:
        if p[0] == "r":
            out += bytes([p[2] - 3, p[1]])
            dec += bytes([p[1]]) * p[2]
        else:
            out += bytes([256 - len(p[1])]) + p[1]
            dec += p[1]
    return bytes(out), bytes(dec)


@pytest.mark.parametrize("shape", ["long", "tiny_between_long", "dense_tiny", "literals_only", "runs_only"])
def test_byte_rle_tile_shapes(gpu, oracle, shape):
    """Plain byte output over run shapes that take every way through the tiled expansion: dwords
    inside one run, across two, across three and more (one- and two-byte literals between runs),
    tiles with more than 64 runs (per-lane walk), odd totals."""
    rng = np.random.default_rng(700 + ["long", "tiny_between_long", "dense_tiny", "literals_only", "runs_only"].index(shape))
    pieces = []
    for i in range(3000):
        def lit(n):
            return ("l", bytes(rng.integers(0, 256, size=n, dtype=np.uint8)))
        def run(n):
            return ("r", int(rng.integers(0, 256)), n)
        if shape == "long":
            pieces.append(run(int(rng.integers(100, 131))) if rng.random() < 0.5 else lit(int(rng.integers(90, 129))))
        elif shape == "tiny_between_long":
            pieces.append(run(130) if i % 2 else lit(128))
            if rng.random() < 0.5:
                pieces += [lit(int(rng.integers(1, 3))), run(3), lit(1)]
        elif shape == "dense_tiny":
            pieces.append(run(int(rng.integers(3, 6))) if rng.random() < 0.5 else lit(int(rng.integers(1, 4))))
        elif shape == "literals_only":
            pieces.append(lit(int(rng.integers(1, 129))))
        else:
            pieces.append(run(int(rng.integers(3, 131))))
    stream, dec = _encode_pieces(pieces)
    want = np.frombuffer(dec, dtype=np.uint8)
    rc, ref = oracle.byte_rle(stream, want.size)
    assert rc == 0 and (ref == want).all()
    for n in (want.size, want.size - 1, want.size - 17, 4096, 4097, 15, 1):
        rc, out = gpu.byte_rle(stream, n)
        assert rc == 0 and (out == want[:n]).all(), n
    # the stream at other 16-byte phases of device memory
    import ctypes as C
    import torch
    from orc_format_amd import _native as N
    from orc_format_amd import default_context
    ctx = default_context()
    for lead in (1, 5, 15):
        d_stream = torch.zeros(lead + len(stream), dtype=torch.uint8, device="cuda")
        d_stream[lead:] = torch.from_numpy(np.frombuffer(stream, np.uint8).copy())
        d_out = torch.zeros(want.size, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        rc = N.lib().orcd_byte_rle_decode(ctx.handle, C.c_void_p(d_stream.data_ptr() + lead), len(stream),
                                          want.size, None, 0, C.c_void_p(d_out.data_ptr()),
                                          N.IN_DEVICE | N.OUT_DEVICE)
        ctx.synchronize()
        assert rc == 0 and (d_out.cpu().numpy() == want).all(), lead


def test_floats(gpu, oracle):
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 256, size=8 * 1001, dtype=np.uint8).tobytes()  # NaN payloads included
    for dtype in (np.float32, np.float64):
        n = len(raw) // np.dtype(dtype).itemsize
        rc, out = gpu.floats(raw, n, dtype)
        rc2, ref = oracle.floats(raw, n, dtype)
        assert rc == 0 and rc2 == 0 and out.tobytes() == ref.tobytes()


def test_strings_random(gpu, oracle):
    rng = np.random.default_rng(4)
    n = 30_000
    lengths = rng.integers(0, 57, size=n)
    alphabet = np.frombuffer("abcdefghijklmnopqrstuvwxyzéü€😀".encode(), dtype=np.uint8)
    text = "".join(rng.choice(list("abcxyz éü€😀"), size=int(lengths.sum())))
    # lengths are in bytes: cut the text at code-point boundaries
    chars = list(text)
    vals, pos = [], 0
    for l in lengths:
        s = ""
        while pos < len(chars) and len((s + chars[pos]).encode()) <= l:
            s += chars[pos]
            pos += 1
        vals.append(s.encode())
    blens = np.array([len(v) for v in vals], dtype=np.uint64)
    data = b"".join(vals)
    # LENGTH stream: Direct runs of width 6
    ls = b"".join(enc.direct(blens[i:i + 512], 6, False) for i in range(0, n, 512))
    rc, off_ref, bytes_ref = oracle.string_direct(ls, data, n)
    assert rc == 0
    rc, off, out = gpu.string_direct(ls, data, n)
    assert rc == 0 and (off == off_ref).all() and out.tobytes() == bytes_ref.tobytes()
    assert off[-1] == len(data)
    # dictionary gather
    idx = rng.integers(0, 500, size=n)
    rc, doff, dbytes = oracle.string_direct(ls, data, 500)
    rc, o_ref, b_ref = oracle.string_dictionary(doff, dbytes, idx)
    rc, o_gpu, b_gpu = gpu.string_dictionary(doff, dbytes, idx)
    assert rc == 0 and (o_gpu == o_ref).all() and b_gpu.tobytes() == b_ref.tobytes()


DICT_SHAPES = ["small_dict", "large_dict", "long_values", "empty_values", "large_short_dict",
               "longest_short", "shortest_long", "one_long_entry"]


@pytest.mark.parametrize("shape", DICT_SHAPES)
def test_dictionary_gather_shapes(gpu, oracle, shape):
    """The gather's ways through: dictionary held in LDS or read from memory, entries all short
    (one kernel: 64 values fit the LDS image) or not (another: the image is filled several times),
    the longest entry on either side of that line, empty entries, every alignment of the output
    stretch."""
    rng = np.random.default_rng(DICT_SHAPES.index(shape) + 40)
    if shape == "small_dict":
        lens = rng.integers(1, 30, size=300)
    elif shape == "large_dict":
        lens = rng.integers(1, 40, size=6000)       # 120 KB of bytes: not in LDS
    elif shape == "long_values":
        lens = rng.integers(0, 400, size=50)         # 64 values often exceed the image
    elif shape == "large_short_dict":
        lens = rng.integers(0, 33, size=8000)        # not in LDS, no entry over 32 bytes
    elif shape == "longest_short":
        lens = rng.integers(28, 33, size=100)        # 64 values fill the image to the last byte
        lens[7] = 32
    elif shape == "shortest_long":
        lens = rng.integers(28, 33, size=100)
        lens[7] = 33
    elif shape == "one_long_entry":
        lens = rng.integers(1, 9, size=400)
        lens[123] = 9000                             # one value spans several fillings of the image
    else:
        lens = rng.integers(0, 3, size=200)          # many empty entries
    doff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    dbytes = rng.integers(0, 256, size=int(doff[-1]), dtype=np.uint8).tobytes()
    # (from 16 384 values on, the sums of a dictionary whose offsets fit LDS come from the table there)
    for n in (1, 63, 64, 65, 1000, 20_001, 300_001):
        idx = rng.integers(0, lens.size, size=n)
        rc, o_ref, b_ref = oracle.string_dictionary(doff, dbytes, idx)
        assert rc == 0
        rc, o_gpu, b_gpu = gpu.string_dictionary(doff, dbytes, idx)
        assert rc == 0 and (o_gpu == o_ref).all() and b_gpu.tobytes() == b_ref.tobytes(), n
    # an index past the dictionary, and offsets that do not describe its bytes, at that size too
    idx = rng.integers(0, lens.size, size=50_000)
    idx[31_337] = lens.size
    assert gpu.string_dictionary(doff, dbytes, idx)[0] == 1
    assert oracle.string_dictionary(doff, dbytes, idx)[0] == 1
    idx[31_337] = 0
    bad = doff.copy()
    if lens.size > 3 and doff[2] < doff[-1]:
        bad[2] = doff[-1] + 5
        assert gpu.string_dictionary(bad, dbytes, idx)[0] != 0


# ---- device pointers and larger sizes ----------------------------------------------------------

def test_device_pointers_and_properties():
    """Device-resident input and output through the raw-pointer entry points, at a size where the
    check is a property: a Direct stream of 17-bit values decodes to the generator's values, and
    a Delta stream decodes to a strictly increasing sequence with the right last value."""
    import torch
    from orc_format_amd import default_context
    ctx = default_context()
    rng = np.random.default_rng(5)
    n_runs = 4096
    n = n_runs * 512
    u = (rng.integers(0, 2**16, size=n, dtype=np.uint64) | np.uint64(1 << 16))
    payload = np.frombuffer(enc.pack_bits(u, 17), dtype=np.uint8).reshape(n_runs, 1088)
    stream = np.concatenate([np.tile(np.array([0x61, 0xFF], dtype=np.uint8), (n_runs, 1)), payload],
                            axis=1).reshape(-1)
    entries = np.arange(0, n_runs, 20, dtype=np.uint64) * np.uint64(1090)
    d_stream = torch.from_numpy(stream.copy()).cuda()
    d_entries = torch.from_numpy(entries.view(np.int64).copy()).cuda()
    d_out = torch.empty(n, dtype=torch.int64, device="cuda")
    ctx.rlev2_device(d_stream.data_ptr(), d_stream.numel(), n, True, d_out.data_ptr(),
                     d_entries.data_ptr(), d_entries.numel())
    ctx.synchronize()
    expect = ((u >> np.uint64(1)).view(np.int64)) ^ -((u & np.uint64(1)).view(np.int64))
    assert (d_out.cpu().numpy() == expect).all()
    assert ctx.phase_ms(3) > 0 and ctx.last_elapsed_ms() > 0

    # Delta: monotonic timestamps, width 10
    d = rng.integers(1, 1000, size=(n_runs, 512), dtype=np.int64)
    t = np.int64(1_600_000_000_000) + np.cumsum(d.reshape(-1))
    chunks = []
    starts = []
    pos = 0
    tt = t.reshape(n_runs, 512)
    for r in range(n_runs):
        buf = enc.delta(int(tt[r, 0]), int(d[r, 1]), d[r, 2:].astype(np.uint64), 10, True)
        starts.append(pos)
        pos += len(buf)
        chunks.append(buf)
    stream = np.frombuffer(b"".join(chunks), dtype=np.uint8)
    d_stream = torch.from_numpy(stream.copy()).cuda()
    d_entries = torch.from_numpy(np.array(starts[::20], dtype=np.int64)).cuda()
    ctx.rlev2_device(d_stream.data_ptr(), d_stream.numel(), n, True, d_out.data_ptr(),
                     d_entries.data_ptr(), d_entries.numel())
    ctx.synchronize()
    got = d_out.cpu().numpy()
    assert (np.diff(got) > 0).all() and got[-1] == t[-1] and (got == t).all()


def test_deserialize_module(golden_dir, expected):
    """orc_format_amd.deserialize: the reference harness's functions (deserialize.rs) over the
    library's own host reader, ROW_INDEX entry points included; checked against pyarrow."""
    orc = pytest.importorskip("pyarrow.orc")
    from orc_format_amd import deserialize as de
    from orc_format_amd import read
    md = read.read_metadata(os.path.join(golden_dir, "test.orc"))
    footer = read.read_stripe_footer(md, 0)
    exp = expected["test.orc"]
    v, x = de.deserialize_int_array(read.read_stripe_column(md, 0, footer, 15))
    assert v.tolist() == exp["validity_15_17"] and x.tolist() == exp["15"]
    v, x = de.deserialize_f32_array(read.read_stripe_column(md, 0, footer, 1))
    assert v.tolist() == exp["validity_1_14"] and x.tolist() == exp["1"]
    v, x = de.deserialize_bool_array(read.read_stripe_column(md, 0, footer, 2))
    assert x.tolist() == exp["2"]
    v, off, data = de.deserialize_str_array(read.read_stripe_column(md, 0, footer, 3))
    assert de.strings(off, data) == exp["3"]
    md = read.read_metadata(os.path.join(golden_dir, "string_dict.orc"))
    footer = read.read_stripe_footer(md, 0)
    v, off, data = de.deserialize_str_array(read.read_stripe_column(md, 0, footer, 1))
    assert de.strings(off, data) == ["abc", "efgh"] * 32

    path = os.path.join(golden_dir, "mixed_compression.orc")
    table = orc.ORCFile(path).read()
    md = read.read_metadata(path)
    footer = read.read_stripe_footer(md, 0)
    for ci, name in enumerate(table.column_names):
        column = read.read_stripe_column(md, 0, footer, ci + 1)
        want = table.column(name).to_pylist()
        kind = md.types[ci + 1]
        if kind in (2, 3, 4):
            v, x = de.deserialize_int_array(column)
            got = assemble.scatter(v, x)
        elif kind == 6:
            v, x = de.deserialize_f64_array(column)
            assert x.tobytes() == np.array([w for w in want if w is not None]).tobytes()
            continue
        elif kind == 0:
            v, x = de.deserialize_bool_array(column)
            got = assemble.scatter(v, x)
        else:
            v, off, data = de.deserialize_str_array(column)
            got = assemble.scatter(v, de.strings(off, data))
        assert got == want, name


def test_dictionary_gather_on_device_with_a_capacity(oracle):
    """Device pointers in and out: the gather is launched before the host knows the column's size,
    so the capacity has to hold by the kernel's own check -- a buffer of exactly the size works, one
    that is too small gives ORCD_INVALID_ARGUMENT and not a byte past its end is written; an index
    past the dictionary is still reported."""
    import torch
    from orc_format_amd import _native as N, default_context
    ctx = default_context()
    lib = N.lib()
    rng = np.random.default_rng(77)
    for max_len in (20, 90):   # both kernels
        lens = rng.integers(0, max_len + 1, size=700)
        doff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        dbytes = rng.integers(0, 256, size=int(doff[-1]), dtype=np.uint8)
        idx = rng.integers(0, lens.size, size=200_003)
        rc, o_ref, b_ref = oracle.string_dictionary(doff, dbytes.tobytes(), idx)
        assert rc == 0
        total = int(o_ref[-1])
        d_off, d_bytes, d_idx = (torch.from_numpy(a).cuda() for a in (doff, dbytes, idx))
        offsets = torch.empty(idx.size + 1, dtype=torch.int64, device="cuda")
        GUARD = 4096
        for cap in (total, total - 1, total // 2, 0):
            out = torch.full((total + GUARD,), 0xA5, dtype=torch.uint8, device="cuda")
            rc = lib.orcd_string_dictionary_decode(ctx.handle, d_off.data_ptr(), d_bytes.data_ptr(), lens.size,
                                                   d_idx.data_ptr(), idx.size, offsets.data_ptr(), out.data_ptr(),
                                                   cap, N.IN_DEVICE | N.OUT_DEVICE)
            torch.cuda.synchronize()
            if cap >= total:
                assert rc == 0
                assert np.array_equal(offsets.cpu().numpy(), o_ref)
                assert out[:total].cpu().numpy().tobytes() == b_ref.tobytes()
            else:
                assert rc == 102, (cap, rc)  # ORCD_INVALID_ARGUMENT
            assert bool((out[cap:] == 0xA5).all()), cap
        bad = idx.copy()
        bad[100_000] = lens.size
        d_bad = torch.from_numpy(bad).cuda()
        out = torch.empty(total + GUARD, dtype=torch.uint8, device="cuda")
        rc = lib.orcd_string_dictionary_decode(ctx.handle, d_off.data_ptr(), d_bytes.data_ptr(), lens.size,
                                               d_bad.data_ptr(), bad.size, offsets.data_ptr(), out.data_ptr(),
                                               out.numel(), N.IN_DEVICE | N.OUT_DEVICE)
        assert rc == 1  # ORCD_OUT_OF_SPEC (deserialize.rs:162)
