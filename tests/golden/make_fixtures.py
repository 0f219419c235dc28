"""Regenerates the ORC fixture files of this directory with pyarrow (Apache ORC C++ writer).

The reference generates its fixtures with `pyorc` (write.py:77-116), which is not installed
here; `pyorc` wraps the same Apache ORC C++ writer, so `pyarrow.orc.write_table` picks the same
RLE v2 sub-encodings for the same rows.  The uncompressed files are produced here, the zlib
ones by make_zlib_fixtures.py.  One known difference: the current writer omits the PRESENT stream of a
column without nulls, where the reference's tests assume it always exists.

Run from the repository root:  python tests/golden/make_fixtures.py
The expected decoded vectors (tests/it/main.rs:26-316) are kept in expected.json.
"""
import json
import os

import numpy as np
import pyarrow as pa
import pyarrow.orc as orc

HERE = os.path.dirname(os.path.abspath(__file__))

# the rows of write.py:5-23, column order preserved (ORC column ids 1..17)
TEST_ROWS = {
    "a": (pa.float32(), [1.0, 2.0, None, 4.0, 5.0]),
    "b": (pa.bool_(), [True, False, None, True, False]),
    "str_direct": (pa.string(), ["a", "cccccc", None, "ddd", "ee"]),
    "d": (pa.string(), ["a", "bb", None, "ccc", "ddd"]),
    "e": (pa.string(), ["ddd", "cc", None, "bb", "a"]),
    "f": (pa.string(), ["aaaaa", "bbbbb", None, "ccccc", "ddddd"]),
    "int_short_repeated": (pa.int32(), [5, 5, None, 5, 5]),
    "int_neg_short_repeated": (pa.int32(), [-5, -5, None, -5, -5]),
    "int_delta": (pa.int32(), [1, 2, None, 4, 5]),
    "int_neg_delta": (pa.int32(), [5, 4, None, 2, 1]),
    "int_direct": (pa.int32(), [1, 6, None, 3, 2]),
    "int_neg_direct": (pa.int32(), [-1, -6, None, -3, -2]),
    "bigint_direct": (pa.int64(), [1, 6, None, 3, 2]),
    "bigint_neg_direct": (pa.int64(), [-1, -6, None, -3, -2]),
    "bigint_other": (pa.int64(), [5, -5, 1, 5, 5]),
    "utf8_increase": (pa.string(), ["a", "bb", "ccc", "dddd", "eeeee"]),
    "utf8_decrease": (pa.string(), ["eeeee", "dddd", "ccc", "bb", "a"]),
}


def _table(columns):
    return pa.table({k: pa.array(v, type=t) for k, (t, v) in columns.items()})


def _write(table, name, **kw):
    path = os.path.join(HERE, name)
    orc.write_table(table, path, compression="uncompressed", **kw)
    # read back with the Apache ORC C++ reader as a sanity check of the file itself
    assert orc.ORCFile(path).read().equals(table), name
    return path


def splitmix64(seed, n):
    """n values of the splitmix64 sequence as uint64 (vectorised)."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def main():
    with np.errstate(over="ignore"):
        _write(_table(TEST_ROWS), "test.orc")
        _write(_table({"long": (pa.bool_(), [True] * 32)}), "long_bool.orc")
        _write(_table({"dict": (pa.string(), ["abcd", "efgh"] * 32)}), "string_long.orc")
        _write(_table({"dict": (pa.string(), ["abc", "efgh"] * 32)}), "string_dict.orc",
               dictionary_key_size_threshold=0.1)
        _write(_table({"dict": (pa.string(), ["abcd", "efgh"] * (10**4 // 2))}),
               "string_long_long.orc")

        # Inputs beyond what the reference can decode; the expected values are whatever
        # pyarrow.orc reads back (checked in _write) and are recomputed in the tests from the
        # same seeds.
        n = 20000
        r = splitmix64(42, n)
        w17 = ((r & np.uint64(0xFFFF)) | np.uint64(0x8000)).astype(np.int64)  # zigzag -> 17 bits
        sorted_ts = np.int64(1_600_000_000_000) + np.cumsum(
            (splitmix64(43, n) % np.uint64(1000) + np.uint64(1)).astype(np.int64))
        outliers = (splitmix64(44, n) % np.uint64(100)).astype(np.int64)
        outliers[::97] += np.int64(1) << np.int64(40)  # forces PatchedBase runs
        lowcard = (splitmix64(45, n) % np.uint64(16)).astype(np.int32)
        lowcard = np.repeat(lowcard[: n // 8], 8)[:n]  # repeats -> ShortRepeat / Delta(0) runs
        nullable = pa.array(
            (splitmix64(46, n) % np.uint64(1 << 20)).astype(np.int64),
            mask=(splitmix64(47, n) % np.uint64(10)) == 0)
        f64 = np.frombuffer(splitmix64(48, n).tobytes(), dtype=np.float64).copy()
        f64[~np.isfinite(f64)] = 0.0
        bools = pa.array((splitmix64(49, n) & np.uint64(1)).astype(bool),
                         mask=(splitmix64(50, n) % np.uint64(7)) == 0)
        strs = pa.array(["s%d" % (int(x) % 1000) * (1 + int(x) % 5)
                         for x in splitmix64(51, n) % np.uint64(10**6)])
        mixed = pa.table({
            "w17": pa.array(w17), "sorted_ts": pa.array(sorted_ts), "outliers": pa.array(outliers),
            "lowcard": pa.array(lowcard), "nullable": nullable, "f64": pa.array(f64),
            "bools": bools, "strs": strs,
        })
        # 'compression' strategy turns off aligned bit packing -> widths such as 17 and 10
        # one large stripe with a row-group index entry every 1000 rows ...
        _write(mixed, "mixed_compression.orc", compression_strategy="compression",
               dictionary_key_size_threshold=0.0, stripe_size=8 * 1024 * 1024,
               row_index_stride=1000, batch_size=20000)
        # ... and many small stripes
        _write(mixed, "mixed_speed.orc", compression_strategy="speed",
               dictionary_key_size_threshold=0.0, stripe_size=256 * 1024, row_index_stride=1000)

    expected = {
        # tests/it/main.rs:26-176, 298-316
        "test.orc": {
            "validity_1_14": [True, True, False, True, True],
            "validity_15_17": [True] * 5,
            "1": [1.0, 2.0, 4.0, 5.0],
            "2": [True, False, True, False],
            "3": ["a", "cccccc", "ddd", "ee"],
            "4": ["a", "bb", "ccc", "ddd"],
            "5": ["ddd", "cc", "bb", "a"],
            "6": ["aaaaa", "bbbbb", "ccccc", "ddddd"],
            "7": [5, 5, 5, 5],
            "8": [-5, -5, -5, -5],
            "9": [1, 2, 4, 5],
            "10": [5, 4, 2, 1],
            "11": [1, 6, 3, 2],
            "12": [-1, -6, -3, -2],
            "13": [1, 6, 3, 2],
            "14": [-1, -6, -3, -2],
            "15": [5, -5, 1, 5, 5],
            "16": ["a", "bb", "ccc", "dddd", "eeeee"],
            "17": ["eeeee", "dddd", "ccc", "bb", "a"],
        },
        # tests/it/main.rs:178-186, 198-230, 249-266
        "long_bool.orc": {"rows": 32, "1": [True] * 32},
        "string_long.orc": {"rows": 64, "1_cycle": ["abcd", "efgh"]},
        "string_dict.orc": {"rows": 64, "1_cycle": ["abc", "efgh"]},
        "string_long_long.orc": {"rows": 10000, "1_cycle": ["abcd", "efgh"]},
    }
    with open(os.path.join(HERE, "expected.json"), "w") as f:
        json.dump(expected, f, indent=1)


if __name__ == "__main__":
    main()
