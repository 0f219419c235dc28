"""The zlib-compressed fixture files (write.py:89, 103, 110, 116 write them with pyorc, which is not
installed here; pyarrow wraps the same Apache ORC C++ writer): the same rows as their uncompressed
twins, compression ZLIB, plus a zlib twin of mixed_compression.orc (read back from it, so the two
hold the same table).  The reference asks for compression blocks of 32 bytes so that chunks end
inside values; this writer only takes multiples of its 64 KiB memory block, so the small files
are one chunk per stream and chunk boundaries inside values are covered by the long float file,
the mixed file and by files from bench_util/orc_write.py, which chunks as finely as asked.

Run from the repository root:  python tests/golden/make_zlib_fixtures.py
"""
import os

import pyarrow as pa
import pyarrow.orc as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def _write(table, name, **kw):
    path = os.path.join(HERE, name)
    orc.write_table(table, path, compression="zlib", **kw)
    assert orc.ORCFile(path).read().equals(table), name
    assert orc.ORCFile(path).compression == "ZLIB"


def main():
    small = dict(compression_block_size=65536)
    _write(pa.table({"long": pa.array([True] * 32, type=pa.bool_())}), "long_bool_gzip.orc", **small)
    _write(pa.table({"dict": pa.array(["abc", "efgh"] * 32)}), "string_dict_gzip.orc",
           dictionary_key_size_threshold=0.1, **small)
    _write(pa.table({"dict": pa.array(["abcd", "efgh"] * (10**4 // 2))}), "string_long_long_gzip.orc", **small)
    _write(pa.table({"dict": pa.array([1.0, 2.0, 4.0, 5.0] * (10**6 // 4), type=pa.float32())}),
           "f32_long_long_gzip.orc", compression_block_size=65536)
    mixed = orc.ORCFile(os.path.join(HERE, "mixed_compression.orc")).read()
    _write(mixed, "mixed_zlib.orc", compression_strategy="compression", dictionary_key_size_threshold=0.0,
           stripe_size=256 * 1024, row_index_stride=1000, compression_block_size=65536)


if __name__ == "__main__":
    main()
