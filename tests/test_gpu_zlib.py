"""CompressionKind ZLIB (src/read/decompress/mod.rs:9-93), and Snappy, LZ4 and Zstandard behind the
same chunk framing: the host decompresses the chunks, the GPU decodes the same streams as ever.  The reference's compressed tests
(tests/it/main.rs:189-196, 233-247, 269-285, 288-296), a zlib twin of the mixed fixture against its
uncompressed original and pyarrow, and a file whose chunks end inside values."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["host", "device", "both"])
def inflate_where(request, monkeypatch):
    """Every test three times: the chunks inflated by the host's zlib, by the GPU (which by itself
    takes over only when a stripe holds many chunks), and every other chunk by each (by itself
    the library gives the host the chunks the GPU would take longest over)."""
    if request.param == "host":
        monkeypatch.setenv("ORCD_GPU_INFLATE", "0")
    else:
        monkeypatch.setenv("ORCD_GPU_INFLATE_MIN_CHUNKS", "1")
        monkeypatch.setenv("ORCD_GPU_INFLATE_SPLIT", "0" if request.param == "device" else "2")
    return request.param


@pytest.fixture(scope="module")
def ctx():
    from orc_format_amd import default_context
    return default_context()


def column(ctx, path, col, stripe=0, **kw):
    from orc_format_amd import read
    md = read.read_metadata(path)
    footer = read.read_stripe_footer(md, stripe)
    return ctx.decode_column(md, stripe, footer, col, **kw)


def strings_of(c):
    b = c["bytes"].tobytes()
    o = c["offsets"]
    return [b[o[i]:o[i + 1]].decode() for i in range(len(o) - 1)]


def test_read_bool_compressed(ctx, golden_dir):
    c = column(ctx, os.path.join(golden_dir, "long_bool_gzip.orc"), 1)
    assert c["validity"].tolist() == [True] * 32 and c["values"].astype(bool).tolist() == [True] * 32


def test_read_string_dict_gzip(ctx, golden_dir):
    c = column(ctx, os.path.join(golden_dir, "string_dict_gzip.orc"), 1)
    assert c["validity"].tolist() == [True] * 64
    assert strings_of(c) == ["abc", "efgh"] * 32


def test_read_string_long_long_gzip(ctx, golden_dir):
    c = column(ctx, os.path.join(golden_dir, "string_long_long_gzip.orc"), 1)
    assert c["validity"].tolist() == [True] * 10000
    assert strings_of(c) == ["abcd", "efgh"] * 5000


def test_read_f32_long_long_gzip(ctx, golden_dir):
    """main.rs:288-296 only checks the lengths (its values are random); these are written here."""
    c = column(ctx, os.path.join(golden_dir, "f32_long_long_gzip.orc"), 1)
    assert c["validity"].size == 10**6 and c["validity"].all()
    assert c["values"].dtype == np.float32
    assert (c["values"] == np.tile(np.array([1.0, 2.0, 4.0, 5.0], dtype=np.float32), 250000)).all()


def test_mixed_zlib_equals_uncompressed_and_pyarrow(ctx, golden_dir):
    po = pytest.importorskip("pyarrow.orc")
    from orc_format_amd import read
    plain = os.path.join(golden_dir, "mixed_compression.orc")
    packed = os.path.join(golden_dir, "mixed_zlib.orc")
    table = po.ORCFile(packed).read()
    md = read.read_metadata(packed)
    n_cols = len(md.types) - 1
    got = [[] for _ in range(n_cols)]
    for s in range(len(md.stripes)):
        footer = read.read_stripe_footer(md, s)
        cols = ctx.decode_stripe(md, s, footer, list(range(1, n_cols + 1)), row_aligned=True)
        for k, c in enumerate(cols):
            got[k].append(c)
    md0 = read.read_metadata(plain)
    ref = ctx.decode_stripe(md0, 0, read.read_stripe_footer(md0, 0), list(range(1, n_cols + 1)), row_aligned=True)
    for k in range(n_cols):
        valid = np.concatenate([c["validity"] for c in got[k]])
        assert (valid == ref[k]["validity"]).all()
        expect = table.column(k).to_pylist()
        assert valid.tolist() == [x is not None for x in expect]
        if "values" in ref[k]:
            vals = np.concatenate([c["values"] for c in got[k]])
            assert vals.dtype == ref[k]["values"].dtype
            assert (vals.view(np.uint8) == ref[k]["values"].view(np.uint8)).all()
        else:
            strs = sum((strings_of(c) for c in got[k]), [])
            assert strs == strings_of(ref[k])
            assert [s for s, v in zip(strs, valid) if v] == [x for x in expect if x is not None]


def test_chunks_that_end_inside_values(ctx, tmp_path):
    """A file of 97-byte compression chunks (bench_util/orc_write.py), decoded stripe by stripe with
    the columns left on the GPU, against what was written."""
    from bench_util import orc_write
    from orc_format_amd import Context, read, shard
    path = str(tmp_path / "z.orc")
    templates, _ = orc_write.write_string_file(path, 5, 3000, seed=9, compression_block=97)
    md = read.read_metadata(path)
    assert int(md.compression) == 1
    extra = Context(0)

    def on_stripe(context, stripe, cols):
        col = cols[0]
        out = (context.copy_to_host(col.offsets_ptr, col.num_slots + 1, np.int64),
               context.copy_to_host(col.bytes_ptr, col.num_bytes, np.uint8))
        col.free()
        return out
    res = shard.decode_file_shard(ctx, md, 0, 1, columns=[1], row_aligned=False, extra_contexts=[extra],
                                  device=True, on_stripe=on_stripe)
    for stripe, _, (offsets, data) in res:
        lengths, want = templates[stripe % len(templates)]
        assert (np.diff(offsets) == lengths.astype(np.int64)).all() and (data == want).all()
    extra.close()


@pytest.mark.parametrize("codec", ["zlib", "snappy", "lz4", "zstd"])
def test_every_codec_decodes_to_what_pyarrow_reads(tmp_path, codec):
    """A multi-stripe file with nulls, sorted keys, low-cardinality values, dictionary and direct
    strings, written with each codec: the stripes decoded on the GPU (the host decompresses the
    chunks, decompress/mod.rs:60-80; Snappy, LZ4 and Zstandard go beyond the reference) are the
    table pyarrow reads."""
    pa = pytest.importorskip("pyarrow")
    orc = pytest.importorskip("pyarrow.orc")
    from orc_format_amd import Context, read, shard
    rng = np.random.default_rng(41)
    rows = 200_000
    table = pa.table({
        "k": pa.array(np.cumsum(rng.integers(1, 90, size=rows))),
        "v": pa.array(rng.integers(-2**40, 2**40, size=rows), mask=rng.random(rows) < 0.15),
        "c": pa.array(rng.integers(0, 8, size=rows).astype(np.int32)),
        "d": pa.array(["name-%d" % k for k in rng.integers(0, 300, size=rows)]),
        "u": pa.array(["%x" % k for k in rng.integers(0, 2**60, size=rows)]),
        "f": pa.array(rng.standard_normal(rows)),
        "b": pa.array(rng.random(rows) < 0.3),
    })
    path = str(tmp_path / ("t_%s.orc" % codec))
    orc.write_table(table, path, compression=codec, compression_block_size=65536, stripe_size=2 << 20)
    md = read.read_metadata(path)
    want = orc.ORCFile(path).read()
    with Context(0) as ctx:
        pieces = shard.decode_file_shard(ctx, md, 0, 1, row_aligned=False)
    assert len(pieces) == len(md.stripes) >= 2
    first = 0
    for stripe, row0, cols in pieces:
        assert row0 == first
        n = cols[0]["validity"].size
        part = want.slice(first, n)
        for ci, name in enumerate(table.column_names):
            col, ref = cols[ci], part.column(name)
            assert col["validity"].tolist() == ref.is_valid().to_pylist(), (codec, stripe, name)
            dense = ref.drop_null()
            if "offsets" in col:
                raw, off = col["bytes"].tobytes(), col["offsets"]
                got = [raw[off[i]:off[i + 1]].decode() for i in range(len(off) - 1)]
                assert got == dense.to_pylist(), (codec, stripe, name)
            elif name == "f":
                assert np.array_equal(col["values"], np.array(dense.to_pylist())), (codec, stripe, name)
            elif name == "b":
                assert col["values"].astype(bool).tolist() == dense.to_pylist(), (codec, stripe, name)
            else:
                assert (col["values"] == np.array(dense.to_pylist(), dtype=np.int64)).all(), (codec, stripe, name)
        first += n
    assert first == rows


def test_a_compressed_stripe_read_ahead_decodes_the_same(tmp_path, inflate_where):
    """orcd_stripe_prefetch on a zlib file: the stripe's bytes wait in host memory (for the walk over
    the chunk headers) and on the device (for the inflate); what is decoded is what a decode
    without the hint gives, whether the hint was for these columns, for others, or never used."""
    pa = pytest.importorskip("pyarrow")
    orc = pytest.importorskip("pyarrow.orc")
    from orc_format_amd import Context, read
    rng = np.random.default_rng(17)
    rows = 120_000
    table = pa.table({
        "a": pa.array(np.cumsum(rng.integers(-5, 9, size=rows))),
        "s": pa.array(["w%d" % k for k in rng.integers(0, 5000, size=rows)], mask=rng.random(rows) < 0.1),
        "d": pa.array(rng.standard_normal(rows)),
    })
    path = str(tmp_path / "ahead.orc")
    orc.write_table(table, path, compression="zlib", compression_block_size=65536, stripe_size=1 << 19)
    md = read.read_metadata(path)
    n = len(md.stripes)
    footers = [read.read_stripe_footer(md, si) for si in range(n)]

    def plain(cols):
        return [{k: (v.tobytes() if hasattr(v, "tobytes") else v) for k, v in c.items()} for c in cols]
    with Context(0) as ctx:
        want = [plain(ctx.decode_stripe(md, si, footers[si], [1, 2, 3])) for si in range(n)]
    with Context(0) as ctx:
        for si in range(n):
            if si + 1 < n:
                ctx.stripe_prefetch(md, si + 1, footers[si + 1], [1, 2, 3])
            assert plain(ctx.decode_stripe(md, si, footers[si], [1, 2, 3])) == want[si], si
        # a hint for other columns than are decoded, and one that is never used
        ctx.stripe_prefetch(md, 0, footers[0], [1, 3])
        assert plain(ctx.decode_stripe(md, 0, footers[0], [1, 2, 3])) == want[0]
        ctx.stripe_prefetch(md, n - 1, footers[n - 1], [2])
