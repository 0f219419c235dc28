"""Decode rate of a zlib-compressed file, chunks inflated by the host's threads against the GPU.

    python bench_util/zlib_file.py [rows]

Writes (pyarrow) a file of four columns -- slowly varying and random 64-bit integers, doubles,
short strings -- with zlib at 64 KiB and 256 KiB compression blocks, and decodes every stripe of it
through orcd_decode_stripe, results left on the device."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import pyarrow as pa  # noqa: E402
import pyarrow.orc as orc  # noqa: E402


def decode_all(path, repeat=2, ahead=False):
    from orc_format_amd import Context, read
    md = read.read_metadata(path)
    footers = [read.read_stripe_footer(md, si) for si in range(len(md.stripes))]
    best = 1e9
    with Context(0) as ctx:
        for _ in range(repeat + 1):   # (the first pass warms the context's blocks)
            t0 = time.perf_counter()
            for si in range(len(md.stripes)):
                if ahead and si + 1 < len(md.stripes):
                    ctx.stripe_prefetch(md, si + 1, footers[si + 1], [1, 2, 3, 4])
                cols = ctx.decode_stripe(md, si, footers[si], [1, 2, 3, 4], device=True)
                del cols
            ctx.synchronize()
            best = min(best, time.perf_counter() - t0)
    return best, len(md.stripes)


def decode_all_contexts(path, n_contexts, repeat=2):
    """The same through decode_file_shard with several contexts on the GPU, a host thread each: one
    stripe's read and chunk walk overlap another's inflate."""
    from orc_format_amd import Context, read, shard
    md = read.read_metadata(path)
    contexts = [Context(0) for _ in range(n_contexts)]

    def on_stripe(context, stripe, cols):
        for c in cols:
            c.free()
        return None
    best = 1e9
    for _ in range(repeat + 1):
        t0 = time.perf_counter()
        shard.decode_file_shard(contexts[0], md, 0, 1, columns=[1, 2, 3, 4], row_aligned=False,
                                extra_contexts=contexts[1:], device=True, on_stripe=on_stripe)
        for c in contexts:
            c.synchronize()
        best = min(best, time.perf_counter() - t0)
    for c in contexts:
        c.close()
    return best


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    rng = np.random.default_rng(5)
    table = pa.table({
        "walk": pa.array(np.cumsum(rng.integers(-50, 51, size=rows))),
        "wide": pa.array(rng.integers(0, 2**40, size=rows)),
        "real": pa.array(np.round(rng.standard_normal(rows), 3)),
        "word": pa.array(np.char.add("k", rng.integers(0, 10**6, size=rows).astype(str))),
    })
    plain = table.nbytes
    with tempfile.TemporaryDirectory() as tmp:
        for block in (1 << 16, 1 << 18):
            path = os.path.join(tmp, "z%d.orc" % block)
            orc.write_table(table, path, compression="zlib", compression_block_size=block, stripe_size=64 << 20,
                            dictionary_key_size_threshold=0.0)
            size = os.path.getsize(path)
            out = {}
            for where in ("host", "device"):
                os.environ.pop("ORCD_GPU_INFLATE", None)
                if where == "host":
                    os.environ["ORCD_GPU_INFLATE"] = "0"
                dt, stripes = decode_all(path)
                out[where] = dt
            out["ahead"], _ = decode_all(path, ahead=True)
            os.environ.pop("ORCD_GPU_INFLATE", None)
            several = {n: decode_all_contexts(path, n) for n in (2, 4)}
            print("  device inflate, contexts: " + ", ".join("%d: %.3f s (%.2f GB/s of file)" % (n, dt, size / dt / 1e9)
                                                           for n, dt in several.items()), flush=True)
            os.environ["ORCD_GPU_INFLATE"] = "0"
            several = {n: decode_all_contexts(path, n) for n in (2, 4)}
            os.environ.pop("ORCD_GPU_INFLATE", None)
            print("  host inflate, contexts: " + ", ".join("%d: %.3f s (%.2f GB/s of file)" % (n, dt, size / dt / 1e9)
                                                         for n, dt in several.items()), flush=True)
            print("zlib file, block %d KiB: %d rows, %.0f MB on disk, %.0f MB as arrays, %d stripes: host inflate "
                  "%.3f s (%.2f GB/s of file), device inflate %.3f s (%.2f GB/s of file), "
                  "with the next stripe read ahead %.3f s (%.2f GB/s)" % (
                      block >> 10, rows, size / 1e6, plain / 1e6, stripes, out["host"], size / out["host"] / 1e9,
                      out["device"], size / out["device"] / 1e9, out["ahead"], size / out["ahead"] / 1e9), flush=True)


if __name__ == "__main__":
    main()
