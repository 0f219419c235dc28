"""The host reader (file tail, stripe footers, column regions, ROW_INDEX) on damaged files: any
status is fine, a crash or a hang is not.  No GPU involved."""
import ctypes as C
import os

import numpy as np
import pytest

from orc_format_amd import _native as N

FILES = ["test.orc", "string_dict.orc", "mixed_compression.orc", "mixed_speed.orc",
         "string_dict_gzip.orc", "mixed_zlib.orc"]   # (compression chunks: damaged deflate streams and headers)


def _walk(lib, image):
    """Opens the image and touches everything the reader offers; returns the statuses seen."""
    seen = []
    buf = np.frombuffer(image, dtype=np.uint8)
    h = C.c_void_p()
    rc = lib.orcd_file_open_memory(buf.ctypes.data, buf.size, C.byref(h))
    seen.append(rc)
    if rc != 0:
        return seen
    try:
        n_types = lib.orcd_file_num_types(h)
        lib.orcd_file_number_of_rows(h)
        lib.orcd_file_row_index_stride(h)
        lib.orcd_file_compression(h)
        for t in range(min(n_types, 64)):
            lib.orcd_file_type_kind(h, t)
        for stripe in range(min(lib.orcd_file_num_stripes(h), 8)):
            si = N.StripeInfo()
            seen.append(lib.orcd_file_stripe(h, stripe, C.byref(si)))
            ft = C.c_void_p()
            rc = lib.orcd_read_stripe_footer(h, stripe, C.byref(ft))
            seen.append(rc)
            if rc != 0:
                continue
            for i in range(min(lib.orcd_stripe_footer_num_streams(ft), 256)):
                info = N.StreamInfo()
                lib.orcd_stripe_footer_stream(ft, i, C.byref(info))
            for column in range(min(n_types, 24) + 2):
                col = C.c_void_p()
                rc = lib.orcd_read_stripe_column(h, stripe, ft, column, C.byref(col))
                seen.append(rc)
                if rc != 0:
                    continue
                kind, size = C.c_int(), C.c_uint32()
                lib.orcd_column_encoding(col, C.byref(kind), C.byref(size))
                lib.orcd_column_number_of_rows(col)
                for k in range(8):
                    ptr, ln = C.c_void_p(), C.c_size_t()
                    if lib.orcd_column_get_stream(col, k, C.byref(ptr), C.byref(ln)) == 0 and ln.value:
                        # the view must lie inside readable memory: touch both ends
                        view = (C.c_uint8 * ln.value).from_address(ptr.value)
                        _ = view[0] + view[ln.value - 1]
                    ent, n = C.c_void_p(), C.c_size_t()
                    if lib.orcd_column_entry_points(h, col, k, C.byref(ent), C.byref(n)) == 0 and n.value:
                        arr = (C.c_uint64 * n.value).from_address(ent.value)
                        _ = arr[0] + arr[n.value - 1]
                lib.orcd_column_free(col)
            lib.orcd_stripe_footer_free(ft)
    finally:
        lib.orcd_file_close(h)
    return seen


@pytest.mark.parametrize("name", FILES)
def test_damaged_files_do_not_crash_the_reader(golden_dir, name):
    lib = N.lib()
    image = open(os.path.join(golden_dir, name), "rb").read()
    assert all(rc == 0 for rc in _walk(lib, image) if rc in (0,)) and _walk(lib, image)[0] == 0
    rng = np.random.default_rng(hash(name) % 1000)
    n = len(image)
    opened = failed = 0
    for case in range(300):
        m = bytearray(image)
        kind = case % 5
        if kind == 0:    # flips anywhere
            for _ in range(int(rng.integers(1, 6))):
                m[int(rng.integers(0, n))] = int(rng.integers(0, 256))
        elif kind == 1:  # flips in the tail (postscript, footer, metadata)
            for _ in range(int(rng.integers(1, 6))):
                m[n - 1 - int(rng.integers(0, min(n, 600)))] = int(rng.integers(0, 256))
        elif kind == 2:  # truncation
            m = m[:int(rng.integers(0, n))]
        elif kind == 3:  # the tail glued onto a shorter body
            cut = int(rng.integers(0, n // 2 + 1))
            m = m[:cut] + m[n - int(rng.integers(1, min(n, 400))):]
        else:            # a run of 0xff (maximal varints) somewhere in the tail
            p = n - 1 - int(rng.integers(0, min(n, 500)))
            m[p:p + int(rng.integers(1, 12))] = b"\xff" * 11
        seen = _walk(lib, bytes(m))
        if seen[0] == 0:
            opened += 1
        else:
            failed += 1
    assert opened > 0 and failed > 0


def test_empty_and_tiny_images():
    lib = N.lib()
    for image in (b"", b"O", b"ORC", b"ORC\x00", b"\x00" * 64, b"ORC" + b"\xff" * 40):
        seen = _walk(lib, image)
        assert seen[0] != 0


def test_reader_under_address_and_undefined_behaviour_sanitizers(golden_dir, tmp_path):
    """The same walk as a standalone program built from orcd_host.cpp with
    -fsanitize=address,undefined (host code only): 1600 damaged files, no report."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_fuzz")
    build = subprocess.run(
        [gxx, "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
         "-static-libasan", "-I", os.path.join(root, "include"),
         os.path.join(root, "orc_format_amd", "csrc", "orcd_host.cpp"),
         os.path.join(root, "tests", "host_fuzz_driver.cpp"), "-o", exe, "-lz", "-ldl", "-pthread"],
        capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and "asan" in build.stderr.lower():
        pytest.skip("the sanitizer runtime is not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    files = [os.path.join(golden_dir, f) for f in FILES]
    # (files of the other codecs, written here: damaged Snappy elements, LZ4 blocks and Zstandard frames)
    try:
        import pyarrow as pa
        import pyarrow.orc as orc
        rng = np.random.default_rng(2)
        table = pa.table({"a": pa.array(np.cumsum(rng.integers(1, 9, size=3000))),
                          "s": pa.array(["w%d" % k for k in rng.integers(0, 40, size=3000)])})
        for codec in ("snappy", "lz4", "zstd"):
            path = str(tmp_path / ("fuzz_%s.orc" % codec))
            orc.write_table(table, path, compression=codec, compression_block_size=65536)
            files.append(path)
    except ImportError:
        pass
    # once the plain way, once with every compressed stream inflated by several threads
    for env in ({}, {"ORCD_INFLATE_PARALLEL_MIN": "1"}):
        run = subprocess.run([exe] + files, capture_output=True, text=True,
                             timeout=300, env=dict(os.environ, **env))
        assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
        assert "opened" in run.stdout
