"""Experiment: a real ORC file (written here by the Apache ORC C++ writer through pyarrow:
nullable int64 columns, a double and a string column, uncompressed) read three ways: pyarrow's
own reader on the host, this library column by column, and this library one stripe per call
(PRESENT and integer DATA streams batched).  Results stay in device memory for the two GPU ways;
times include reading the file."""
import ctypes as C
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import pyarrow as pa  # noqa: E402
import pyarrow.orc as orc  # noqa: E402
import torch  # noqa: E402,F401

from orc_format_amd import _native as N  # noqa: E402
from orc_format_amd import Context, read  # noqa: E402
from orc_format_amd.error import check  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    n_int = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    rng = np.random.default_rng(1)
    cols = {}
    for i in range(n_int):
        v = rng.integers(-2**40, 2**40, size=rows) if i % 3 else np.cumsum(rng.integers(1, 1000, size=rows))
        cols["i%d" % i] = pa.array(v, mask=rng.random(rows) < 0.1)
    cols["f"] = pa.array(rng.standard_normal(rows))
    cols["s"] = pa.array(["value-%d" % k for k in rng.integers(0, 10**9, size=rows)])
    path = os.path.join(tempfile.mkdtemp(), "wide.orc")
    orc.write_table(pa.table(cols), path, compression="uncompressed", stripe_size=256 << 20)
    size = os.path.getsize(path)
    print("file: %d rows x %d columns, %.1f MB" % (rows, len(cols), size / 1e6), flush=True)

    def best_of(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best

    want = orc.ORCFile(path).read()
    t_arrow = best_of(lambda: orc.ORCFile(path).read())
    ctx = Context(0)
    lib = N.lib()

    def gpu(per_stripe, out_flag=N.OUT_DEVICE):
        md = read.read_metadata(path)
        for stripe in range(len(md.stripes)):
            footer = read.read_stripe_footer(md, stripe)
            columns = list(range(1, len(md.types)))
            if per_stripe:
                arr = (C.c_uint32 * len(columns))(*columns)
                cds = (N.ColumnData * len(columns))()
                check(lib.orcd_decode_stripe(ctx.handle, md._h, stripe, footer._h, arr, len(columns),
                                             out_flag | N.CHECK_UTF8, cds), ctx.handle)
                for cd in cds:
                    lib.orcd_column_data_free(C.byref(cd))
            else:
                for col in columns:
                    cd = N.ColumnData()
                    check(lib.orcd_decode_column(ctx.handle, md._h, stripe, footer._h, col,
                                                 N.OUT_DEVICE | N.CHECK_UTF8, C.byref(cd)), ctx.handle)
                    lib.orcd_column_data_free(C.byref(cd))
        ctx.synchronize()

    # parity of one stripe against pyarrow before timing
    md = read.read_metadata(path)
    footer = read.read_stripe_footer(md, 0)
    got = ctx.decode_stripe(md, 0, footer, [1, 2], row_aligned=True)
    n0 = md.stripes[0].number_of_rows
    for k, g in enumerate(got):
        w = want.column(k).slice(0, n0)
        assert g["validity"].tolist() == w.is_valid().to_pylist()
        assert (g["values"][g["validity"]] == np.array(w.drop_null().to_pylist())).all()
    t_col = best_of(lambda: gpu(False))
    t_stripe = best_of(lambda: gpu(True))

    # stripes dealt to two host threads, each with a context (and stream) of its own: one
    # stripe's read and copy overlap the other's decode
    from concurrent.futures import ThreadPoolExecutor
    ctxs = [ctx, Context(0)]

    def one_stripe(args):
        k, md, stripe = args
        footer = read.read_stripe_footer(md, stripe)
        columns = list(range(1, len(md.types)))
        arr = (C.c_uint32 * len(columns))(*columns)
        cds = (N.ColumnData * len(columns))()
        check(lib.orcd_decode_stripe(ctxs[k].handle, md._h, stripe, footer._h, arr, len(columns),
                                     N.OUT_DEVICE | N.CHECK_UTF8, cds), ctxs[k].handle)
        for cd in cds:
            lib.orcd_column_data_free(C.byref(cd))

    def gpu_two():
        md = read.read_metadata(path)
        with ThreadPoolExecutor(2) as pool:
            list(pool.map(one_stripe, [(s % 2, md, s) for s in range(len(md.stripes))]))
        for c in ctxs:
            c.synchronize()

    t_two = best_of(gpu_two)
    t_host = best_of(lambda: gpu(True, 0))   # results copied back into malloc'd host arrays
    for name, t in (("pyarrow (host)", t_arrow), ("per column", t_col), ("per stripe", t_stripe),
                    ("2 contexts", t_two), ("stripe -> host", t_host)):
        print("%-15s %.1f ms  %.2f GB/s of file" % (name, t * 1e3, size / t / 1e9), flush=True)


if __name__ == "__main__":
    main()
