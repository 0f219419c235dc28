"""Compression chunks inflated on the GPU (orcd_inflate_chunks) against zlib: the crate hands each
chunk to flate2 (src/read/decompress/mod.rs:60-80); Python's zlib module is the same library and
is the checker here -- same bytes, and the same chunks refused."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from orc_format_amd import default_context
    return default_context()


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    return c.compress(data) + c.flush()


def cpu_inflate(raw, cap):
    """(ok, bytes) the way the host path takes a chunk: a complete raw deflate stream of at most
    `cap` bytes; trailing input is ignored."""
    d = zlib.decompressobj(-15)
    try:
        out = d.decompress(raw, cap + 1)
    except zlib.error:
        return False, b""
    if len(out) > cap or not d.eof:
        return False, b""
    return True, out


def run(ctx, raws, cap):
    """Inflates the chunks `raws` in one call, packed back to back with odd gaps."""
    from orc_format_amd.error import OrcError
    src = bytearray()
    chunks = []
    for k, r in enumerate(raws):
        src += b"\x5a" * (k % 5)   # every alignment
        chunks.append((len(src), len(r), k * (cap + 3), cap))
        src += r
    try:
        out, sizes = ctx.inflate_chunks(bytes(src) if src else b"\0", chunks, len(raws) * (cap + 3))
    except OrcError as e:
        return e.code, None
    return 0, [out[k * (cap + 3):k * (cap + 3) + int(sizes[k])].tobytes() for k in range(len(raws))]


def samples(rng):
    text = (b"the quick brown fox jumps over the lazy dog. " * 400)
    yield b""
    yield b"a"
    yield b"abc" * 5
    yield bytes(1000)
    yield text
    yield rng.integers(0, 256, size=70_000, dtype=np.uint8).tobytes()            # stored blocks
    yield rng.integers(0, 4, size=200_000, dtype=np.uint8).tobytes()              # short codes
    yield np.cumsum(rng.integers(-2, 3, size=60_000)).astype(np.int64).tobytes()  # integer columns
    yield rng.integers(0, 2**40, size=30_000).astype(np.int64).tobytes()
    yield bytes(rng.integers(97, 123, size=1 << 18, dtype=np.uint8)) + text       # > the window
    yield (b"x" * 100_000) + bytes(rng.integers(0, 256, size=3000, dtype=np.uint8)) * 30   # long distances


def test_streams_of_every_kind_decompress_to_what_zlib_gives(ctx):
    rng = np.random.default_rng(1)
    raws, want = [], []
    for data in samples(rng):
        for level, strategy in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (0, zlib.Z_DEFAULT_STRATEGY)):
            raws.append(deflate(data, level, strategy))
            want.append(data)
    cap = max(len(w) for w in want)
    rc, got = run(ctx, raws, cap)
    assert rc == 0
    for k in range(len(raws)):
        assert got[k] == want[k], k
    # trailing bytes after the last block are ignored, as by the host path
    rc, got = run(ctx, [r + b"\xff\x00\x12" for r in raws[:20]], cap)
    assert rc == 0 and got == want[:20]


def test_room_and_truncation(ctx):
    data = b"hello world, " * 1000
    raw = deflate(data)
    assert run(ctx, [raw], len(data))[0] == 0
    assert run(ctx, [raw], len(data) - 1)[0] == 6         # ORCD_DECOMPRESSION: more than the room
    for cut in (1, 2, len(raw) // 2, len(raw) - 1):
        ok, _ = cpu_inflate(raw[:cut], len(data))
        assert not ok
        assert run(ctx, [raw[:cut]], len(data))[0] == 6, cut
    stored = deflate(data, 0)
    assert run(ctx, [stored[:-1]], len(data))[0] == 6
    assert run(ctx, [stored], len(data))[1] == [data]


def test_damaged_chunks_are_refused_exactly_when_zlib_refuses_them(ctx):
    rng = np.random.default_rng(2)
    base = []
    for data in list(samples(rng))[1:9]:
        data = data[:20_000]
        for level, strategy in ((6, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY)):
            base.append(deflate(data, level, strategy))
    cap = 1 << 16
    ok = bad = 0
    for round_ in range(6):
        raws = []
        for raw in base:
            m = bytearray(raw)
            kind = int(rng.integers(0, 4))
            if kind == 0:      # a bit flipped, early (headers, code lengths) or anywhere
                at = int(rng.integers(0, min(len(m), 40 if round_ % 2 else len(m))))
                m[at] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:    # a byte replaced
                m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
            elif kind == 2:    # cut short
                m = m[:int(rng.integers(0, len(m)))]
            else:              # bytes dropped from the middle
                at = int(rng.integers(0, len(m)))
                del m[at:at + int(rng.integers(1, 4))]
            raws.append(bytes(m) if m else b"\x03")
        # one call per chunk: a call reports the failure of any of its chunks
        for raw in raws:
            want_ok, want = cpu_inflate(raw, cap)
            rc, got = run(ctx, [raw], cap)
            assert (rc == 0) == want_ok, (round_, raw[:40].hex(), rc)
            if want_ok:
                assert got[0] == want
                ok += 1
            else:
                assert rc == 6
                bad += 1
    assert ok > 5 and bad > 50


def test_distances_that_reach_the_far_end_of_the_window(ctx):
    """Matches 32 KiB back with literals right behind them: the ring that holds the window wraps
    onto the bytes such a match reads."""
    rng = np.random.default_rng(4)
    raws, want = [], []
    for gap in (32768, 32767, 32760, 32700, 32512, 32768 - 258):
        head = bytes(rng.integers(0, 256, size=400, dtype=np.uint8))
        filler = bytes(rng.integers(0, 256, size=gap - 400, dtype=np.uint8))
        body = bytearray(head + filler)
        for k in range(40):      # pieces of the head again, a fresh byte or two between them
            body += head[k * 9:k * 9 + 3 + k % 7] + bytes(rng.integers(0, 256, size=1 + k % 3, dtype=np.uint8))
        data = bytes(body) * 2
        raws.append(deflate(data, 9))
        want.append(data)
    cap = max(len(w) for w in want)
    rc, got = run(ctx, raws, cap)
    assert rc == 0
    assert got == want


def test_many_chunks_in_one_call(ctx):
    rng = np.random.default_rng(3)
    block = 1 << 16
    col = np.cumsum(rng.integers(-3, 4, size=(1 << 21))).astype(np.int64).tobytes()
    pieces = [col[i:i + block] for i in range(0, len(col), block)]
    raws = [deflate(p) for p in pieces]
    rc, got = run(ctx, raws, block)
    assert rc == 0 and got == pieces


def test_damaged_zlib_files_decode_alike_on_both_paths(tmp_path, monkeypatch):
    """Bytes of a zlib file's stripes damaged at random: the stripe decode with the chunks inflated
    on the GPU, or half of them, ends as the one with the host's zlib does -- the same status, and
    the same arrays when there are any."""
    pa = pytest.importorskip("pyarrow")
    orc = pytest.importorskip("pyarrow.orc")
    from orc_format_amd import Context, read
    from orc_format_amd.error import OrcError
    rng = np.random.default_rng(11)
    rows = 30_000
    table = pa.table({
        "a": pa.array(np.cumsum(rng.integers(-5, 9, size=rows))),
        "b": pa.array(rng.integers(-2**45, 2**45, size=rows), mask=rng.random(rows) < 0.2),
        "s": pa.array(["w%d" % k for k in rng.integers(0, 500, size=rows)]),
        "d": pa.array(rng.standard_normal(rows)),
    })
    path = str(tmp_path / "z.orc")
    orc.write_table(table, path, compression="zlib", compression_block_size=65536, stripe_size=1 << 20)
    image = bytearray(open(path, "rb").read())
    md0 = read.read_metadata(path)
    first, last = int(md0.stripes[0].offset), int(md0.stripes[-1].offset)
    del md0

    def decode(blob, where):
        if where == "host":
            monkeypatch.setenv("ORCD_GPU_INFLATE", "0")
        else:
            monkeypatch.delenv("ORCD_GPU_INFLATE", raising=False)
            monkeypatch.setenv("ORCD_GPU_INFLATE_MIN_CHUNKS", "1")
            monkeypatch.setenv("ORCD_GPU_INFLATE_SPLIT", "0" if where == "device" else "2")
        out = []
        try:
            md = read.read_metadata(blob)
        except OrcError as e:
            return [("open", e.code)]
        with Context(0) as ctx:
            for si in range(len(md.stripes)):
                try:
                    footer = read.read_stripe_footer(md, si)
                    cols = ctx.decode_stripe(md, si, footer, [1, 2, 3, 4])
                except OrcError as e:
                    out.append((si, e.code))
                    continue
                out.append((si, 0, [{k: (v.tobytes() if hasattr(v, "tobytes") else v) for k, v in c.items()}
                                    for c in cols]))
        return out

    same_ok = same_bad = 0
    for trial in range(24):
        blob = bytearray(image)
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(first, max(first + 1, last + 20_000)))
            if at < len(blob):
                blob[at] ^= 1 << int(rng.integers(0, 8))
        a, b, both = decode(bytes(blob), "host"), decode(bytes(blob), "device"), decode(bytes(blob), "both")
        assert len(a) == len(b) == len(both), trial
        for x, y, z in zip(a, b, both):
            assert x[:2] == y[:2] == z[:2], (trial, x[:2], y[:2], z[:2])
            if x[1] == 0 and x[0] != "open":
                assert x[2] == y[2] == z[2], (trial, x[0])
                same_ok += 1
            else:
                same_bad += 1
    assert same_ok > 5 and same_bad > 5
    # the PostScript's compression block size changed under the chunks: one byte too small, tiny,
    # zero, larger than any chunk
    ps = len(image) - 1 - image[-1]
    at = bytes(image).find(b"\x18\x80\x80\x04", ps)
    assert at >= 0
    for claim in (b"\xff\xff\x03", b"\x81\x80\x00", b"\x80\x80\x00", b"\xff\xff\x7f"):
        blob = bytearray(image)
        blob[at + 1:at + 4] = claim
        a, b, both = decode(bytes(blob), "host"), decode(bytes(blob), "device"), decode(bytes(blob), "both")
        assert [x[:2] for x in a] == [x[:2] for x in b] == [x[:2] for x in both], claim
        for x, y, z in zip(a, b, both):
            if x[1] == 0 and x[0] != "open":
                assert x[2] == y[2] == z[2], claim


def test_chunk_tables_that_do_not_fit_are_refused_before_anything_runs(ctx):
    from orc_format_amd.error import OrcError
    raw = deflate(b"abc" * 100)
    for chunks, dst_len in (
        ([(0, len(raw) + 1, 0, 300)], 300),        # more bytes than the source holds
        ([(len(raw) + 5, 1, 0, 300)], 300),        # starts past its end
        ([(0, len(raw), 0, 301)], 300),            # more room than the destination has
        ([(0, len(raw), 400, 10)], 300),           # a place past its end
    ):
        with pytest.raises(OrcError) as e:
            ctx.inflate_chunks(raw, chunks, dst_len)
        assert e.value.code == 102, chunks
    out, sizes = ctx.inflate_chunks(raw, [], 16)   # nothing to do
    assert sizes.size == 0
    assert run(ctx, [b""], 16)[0] == 6             # an empty stream is no deflate stream (zlib: the same)


def test_thousands_of_random_and_half_valid_chunks_in_one_call(ctx):
    """Bytes that are no deflate streams at all, valid streams with their tails replaced by noise,
    and valid ones, 3000 in one launch: every chunk ends as zlib says (sizes[k] is 0xFFFFFFFF for
    the ones that do not inflate), none hangs, and the good ones' bytes are right."""
    import ctypes as C
    from orc_format_amd import _native as N
    rng = np.random.default_rng(21)
    cap = 1 << 15
    good = [deflate(bytes(rng.integers(0, 6, size=int(rng.integers(1, 20_000)), dtype=np.uint8))) for _ in range(40)]
    raws = []
    for k in range(3000):
        kind = k % 3
        if kind == 0:
            raws.append(bytes(rng.integers(0, 256, size=int(rng.integers(1, 3000)), dtype=np.uint8)))
        elif kind == 1:
            g = bytearray(good[k % len(good)])
            at = int(rng.integers(0, len(g)))
            g[at:] = bytes(rng.integers(0, 256, size=len(g) - at, dtype=np.uint8))
            raws.append(bytes(g))
        else:
            raws.append(good[k % len(good)])
    src = b"".join(raws)
    table = np.zeros(len(raws), dtype=[("src_offset", "<u8"), ("dst_offset", "<u8"), ("src_length", "<u4"),
                                       ("dst_capacity", "<u4")])
    at = 0
    for k, r in enumerate(raws):
        table[k] = (at, k * cap, len(r), cap)
        at += len(r)
    s = np.frombuffer(src, dtype=np.uint8)
    out = np.zeros(len(raws) * cap, dtype=np.uint8)
    sizes = np.zeros(len(raws), dtype=np.uint32)
    rc = N.lib().orcd_inflate_chunks(ctx.handle, s.ctypes.data, s.size, table.ctypes.data, len(raws), out.ctypes.data,
                                     out.size, sizes.ctypes.data, 0)
    assert rc in (0, 6)
    ok = bad = 0
    for k, r in enumerate(raws):
        want_ok, want = cpu_inflate(r, cap)
        assert (sizes[k] != 0xFFFFFFFF) == want_ok, k
        if want_ok:
            assert out[k * cap:k * cap + int(sizes[k])].tobytes() == want, k
        ok += want_ok
        bad += not want_ok
    assert rc == (6 if bad else 0)
    assert ok >= 1000 and bad >= 1000
