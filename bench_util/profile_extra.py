"""The kernels that no bench.py workload exercises, one case each, for rocprofv3 (kernel trace and
FETCH_SIZE / WRITE_SIZE passes): a few iterations of one large call, checked once.

    python bench_util/profile_extra.py patched_base | dictionary | byte_rle | scatter | rlev1 | inflate
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from orc_format_amd import _native as N  # noqa: E402
from orc_format_amd import Context  # noqa: E402
from orc_format_amd.error import check  # noqa: E402

ITERS = 4


def timed(ctx, fn):
    best = 1e9
    for _ in range(ITERS):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def patched_base(ctx):
    """2*10^8 values as PatchedBase runs of 512 values, width 8 with 5 % of the values patched to
    24 bits (64 distinct runs, repeated), entry points every 20 runs."""
    import rle_encode as enc
    rng = np.random.default_rng(1)
    runs, vals = [], []
    for _ in range(64):
        v = rng.integers(0, 200, size=512, dtype=np.int64)
        where = np.cumsum(rng.integers(1, 21, size=25)) - 1     # 25 patched positions, gaps of 1..20
        v[where] += rng.integers(1 << 12, 1 << 20, size=25)
        base = int(v.min())
        low = (v - base) & 0xFF
        gaps = np.diff(np.concatenate([[0], where]))
        patches = [(int(g), int((v[i] - base) >> 8)) for g, i in zip(gaps, where)]   # (gap, patch) as stored
        b = enc.patched_base(base, low, 8, patches, 16, 8)
        runs.append(b)
        vals.append(v)
    size = {len(b) for b in runs}
    reps = 2 * 10**8 // (64 * 512)
    stream = torch.from_numpy(np.frombuffer(b"".join(runs), dtype=np.uint8).copy()).cuda().repeat(reps)
    expect = torch.from_numpy(np.concatenate(vals)).cuda().repeat(reps)
    starts = np.cumsum([0] + [len(b) for b in runs] * reps)[:-1]
    entries = torch.from_numpy(starts[::20].astype(np.int64)).cuda()
    n = expect.numel()
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    dt = timed(ctx, lambda: ctx.rlev2_device(stream.data_ptr(), stream.numel(), n, False, out.data_ptr(),
                                             entries.data_ptr(), entries.numel()))
    assert torch.equal(out, expect)
    print("patched_base: %d values, %d stream bytes (run sizes %s), %.3f ms, %.0f GB/s decoded" %
          (n, stream.numel(), sorted(size)[:3], dt * 1e3, n * 8 / dt / 1e9), flush=True)


def offgrid(ctx):
    """The headline column (Direct, width 17, runs of 512) behind one run of 300 values: no run
    starts on a window boundary any more and every window of 512 values holds parts of two runs --
    what a file's column looks like after its first short run.  Entry points every 20 runs."""
    import rle_encode as enc
    from bench_util import synth
    n_runs = int(os.environ.get("OFFGRID_RUNS", str(2 * 10**8 // 512)))
    stream, entries, values = synth.direct_w17(n_runs, seed=42, device="cuda")
    rng = np.random.default_rng(2)
    head_vals = rng.integers(-(1 << 16), 1 << 16, size=300, dtype=np.int64)
    head = np.frombuffer(enc.direct(head_vals, 17, True), dtype=np.uint8).copy()
    stream = torch.cat([torch.from_numpy(head).cuda(), stream])
    expect = torch.cat([torch.from_numpy(head_vals).cuda(), values])
    entries = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), entries[1:] + head.size,
                         ]) if entries.numel() > 1 else torch.zeros(1, dtype=torch.int64, device="cuda")
    n = expect.numel()
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    dt = timed(ctx, lambda: ctx.rlev2_device(stream.data_ptr(), stream.numel(), n, True, out.data_ptr(),
                                             entries.data_ptr(), entries.numel()))
    assert torch.equal(out, expect)
    print("offgrid: %d values, %d stream bytes, %.3f ms, %.0f GB/s decoded, path %d, kernel %s, phases %s" %
          (n, stream.numel(), dt * 1e3, n * 8 / dt / 1e9, ctx.last_path(), ctx.last_value_kernel(),
           [round(ctx.phase_ms(p), 4) for p in (1, 2, 3)]), flush=True)


def dictionary(ctx):
    """10^8 indices into a dictionary of 1000 entries of 4..28 bytes: offsets + gathered bytes.
    (DICT_ENTRIES / DICT_MAX_LEN in the environment: another dictionary, e.g. one too large for LDS.)"""
    import os
    n = 100_000_000
    entries = int(os.environ.get("DICT_ENTRIES", "1000"))
    max_len = int(os.environ.get("DICT_MAX_LEN", "28"))
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    lens = torch.randint(4, max_len + 1, (entries,), device="cuda", generator=g)
    doff = torch.zeros(entries + 1, dtype=torch.int64, device="cuda")
    doff[1:] = torch.cumsum(lens, 0)
    dbytes = torch.randint(97, 123, (int(doff[-1]),), dtype=torch.uint8, device="cuda", generator=g)
    idx = torch.randint(0, entries, (n,), dtype=torch.int64, device="cuda", generator=g)
    want_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    want_off[1:] = torch.cumsum(lens[idx], 0)
    total = int(want_off[-1])
    offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    out = torch.empty(total, dtype=torch.uint8, device="cuda")
    lib = N.lib()
    dt = timed(ctx, lambda: check(lib.orcd_string_dictionary_decode(
        ctx.handle, doff.data_ptr(), dbytes.data_ptr(), entries, idx.data_ptr(), n, offsets.data_ptr(),
        out.data_ptr(), total, N.IN_DEVICE | N.OUT_DEVICE), ctx.handle))
    assert torch.equal(offsets, want_off)
    for k in (0, 1, 12345, n // 2, n - 1):  # a few values, byte for byte
        a, b = int(want_off[k]), int(want_off[k + 1])
        e = int(idx[k])
        assert torch.equal(out[a:b], dbytes[int(doff[e]):int(doff[e + 1])]), k
    probe = torch.arange(0, n, 4099, device="cuda")  # the first byte of every 4099th value
    assert torch.equal(out[want_off[probe]], dbytes[doff[idx[probe]]])
    print("dictionary: %d values, %d bytes out, %.3f ms, %.0f GB/s of bytes + offsets" %
          (n, total, dt * 1e3, (total + 8 * n) / dt / 1e9), flush=True)


def byte_rle(ctx):
    """The byte-RLE mix of the boolean workload, decoded to bytes (orcd_byte_rle_decode): 5*10^8
    bytes out."""
    from bench_util import synth
    stream, entries, packed = synth.byte_rle_mix(500_000_000 // 128, device="cuda")
    n = packed.numel()
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    lib = N.lib()
    dt = timed(ctx, lambda: check(lib.orcd_byte_rle_decode(
        ctx.handle, stream.data_ptr(), stream.numel(), n, entries.data_ptr(), entries.numel(), out.data_ptr(),
        N.IN_DEVICE | N.OUT_DEVICE), ctx.handle))
    assert torch.equal(out, packed)
    print("byte_rle: %d stream bytes -> %d bytes, %.3f ms, %.0f GB/s decoded" %
          (stream.numel(), n, dt * 1e3, n / dt / 1e9), flush=True)


def scatter(ctx):
    """Null scatter: 5*10^8 rows, 10 % nulls, 8-byte values into row positions."""
    rows = 500_000_000
    g = torch.Generator(device="cuda")
    g.manual_seed(2)
    valid = (torch.rand(rows, device="cuda", generator=g) >= 0.1).to(torch.uint8)
    n_values = int(valid.sum())
    dense = torch.arange(n_values, dtype=torch.int64, device="cuda")
    out = torch.empty(rows, dtype=torch.int64, device="cuda")
    lib = N.lib()
    dt = timed(ctx, lambda: check(lib.orcd_scatter_valid(
        ctx.handle, valid.data_ptr(), rows, dense.data_ptr(), n_values, 8, out.data_ptr(),
        N.IN_DEVICE | N.OUT_DEVICE), ctx.handle))
    idx = torch.nonzero(valid[:1_000_000]).flatten()
    assert torch.equal(out[:1_000_000][idx], torch.arange(idx.numel(), dtype=torch.int64, device="cuda"))
    print("scatter: %d rows, %d values, %.3f ms, %.0f GB/s of rows written" %
          (rows, n_values, dt * 1e3, rows * 8 / dt / 1e9), flush=True)


def rlev1(ctx):
    """10^8 values of integer RLE version 1 (a file written as ORC 0.11): 64 distinct row groups of
    10 000 values -- slowly varying values, which the writers turn into a mix of short runs and
    literal groups -- repeated, an entry point per row group."""
    import rle_encode as enc
    rng = np.random.default_rng(3)
    groups, vals = [], []
    for _ in range(64):
        v = np.cumsum(rng.integers(-2, 3, size=10_000)) + int(rng.integers(-10**6, 10**6))
        b, _starts = enc.rlev1(v, True)
        groups.append(b)
        vals.append(v)
    reps = 10**8 // (64 * 10_000)
    stream = torch.from_numpy(np.frombuffer(b"".join(groups), dtype=np.uint8).copy()).cuda().repeat(reps)
    expect = torch.from_numpy(np.concatenate(vals)).cuda().repeat(reps)
    starts = np.cumsum([0] + [len(b) for b in groups] * reps)[:-1]
    entries = torch.from_numpy(starts.astype(np.int64)).cuda()
    n = expect.numel()
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    lib = N.lib()
    dt = timed(ctx, lambda: check(lib.orcd_rlev1_decode(
        ctx.handle, stream.data_ptr(), stream.numel(), n, entries.data_ptr(), entries.numel(), out.data_ptr(),
        N.SIGNED | N.IN_DEVICE | N.OUT_DEVICE), ctx.handle))
    assert torch.equal(out, expect)
    print("rlev1: %d values, %d stream bytes, %d entry points, %.3f ms, %.0f GB/s decoded" %
          (n, stream.numel(), entries.numel(), dt * 1e3, n * 8 / dt / 1e9), flush=True)


def inflate(ctx):
    """Compression chunks of zlib files inflated on the device: 48 MiB each of three kinds of column
    bytes -- slowly varying 32-bit integers, doubles rounded to three decimals, text -- cut into
    chunks of the compression block size, deflated at level 6, the set repeated four times; beside
    it zlib on one host core."""
    import zlib
    rng = np.random.default_rng(4)
    size = 48 << 20
    ints = np.cumsum(rng.integers(-40, 41, size=size // 4)).astype(">i4").tobytes()
    reals = np.round(rng.standard_normal(size // 8) * 100, 3).tobytes()
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 12)), dtype=np.uint8)) for _ in range(5000)]
    text = b" ".join(words[int(k)] for k in rng.integers(0, len(words), size=size // 5))[:size]
    assert len(text) == size
    lib = N.lib()
    for name, data in (("integers", ints), ("doubles", reals), ("text", text)):
        for block in (1 << 16, 1 << 18):
            pieces = [data[i:i + block] for i in range(0, len(data), block)]
            raws = []
            for p in pieces:
                c = zlib.compressobj(6, zlib.DEFLATED, -15)
                raws.append(c.compress(p) + c.flush())
            t0 = time.perf_counter()
            for r in raws[:64]:
                zlib.decompressobj(-15).decompress(r)
            host = sum(len(p) for p in pieces[:64]) / (time.perf_counter() - t0)
            reps = 4
            src = b"".join(raws)
            table = np.zeros(len(raws) * reps, dtype=[("src_offset", "<u8"), ("dst_offset", "<u8"),
                                                      ("src_length", "<u4"), ("dst_capacity", "<u4")])
            k = 0
            for rep in range(reps):
                at = 0
                for r in raws:
                    table[k] = (rep * len(src) + at, k * block, len(r), block)
                    at += len(r)
                    k += 1
            d_src = torch.from_numpy(np.frombuffer(src, dtype=np.uint8).copy()).cuda().repeat(reps)
            d_dst = torch.zeros(len(table) * block, dtype=torch.uint8, device="cuda")
            sizes = np.zeros(len(table), dtype=np.uint32)
            dt = timed(ctx, lambda: check(lib.orcd_inflate_chunks(
                ctx.handle, d_src.data_ptr(), d_src.numel(), table.ctypes.data, len(table), d_dst.data_ptr(),
                d_dst.numel(), sizes.ctypes.data, N.IN_DEVICE | N.OUT_DEVICE), ctx.handle))
            want = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
            total = int(sizes.sum())
            assert total == len(data) * reps
            for rep in range(reps):
                assert torch.equal(d_dst[rep * len(data):(rep + 1) * len(data)], want)
            print("inflate %s: block %d KiB, %d chunks, %.1f MB deflate -> %.1f MB: %.3f ms, %.1f GB/s decompressed "
                  "(zlib on one host core: %.2f GB/s)" % (name, block >> 10, len(table), d_src.numel() / 1e6, total / 1e6,
                                                           dt * 1e3, total / dt / 1e9, host / 1e9), flush=True)


if __name__ == "__main__":
    {"patched_base": patched_base, "offgrid": offgrid, "dictionary": dictionary, "byte_rle": byte_rle, "scatter": scatter,
     "rlev1": rlev1, "inflate": inflate}[sys.argv[1]](Context(0))
