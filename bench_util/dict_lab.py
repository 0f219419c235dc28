"""Experiment: throughput of the dictionary gather (orcd_string_dictionary_decode, device
pointers) on a column of 10^8 values over a dictionary of 1000 entries of 4..28 bytes."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from orc_format_amd import _native as N  # noqa: E402
from orc_format_amd import Context  # noqa: E402
from orc_format_amd.error import check  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    lens = torch.randint(4, 29, (1000,), device="cuda", generator=g)
    doff = torch.zeros(1001, dtype=torch.int64, device="cuda")
    doff[1:] = torch.cumsum(lens, 0)
    dbytes = torch.randint(97, 123, (int(doff[-1]),), dtype=torch.uint8, device="cuda", generator=g)
    idx = torch.randint(0, 1000, (n,), dtype=torch.int64, device="cuda", generator=g)
    want_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    want_off[1:] = torch.cumsum(lens[idx], 0)
    total = int(want_off[-1])
    offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    out = torch.empty(total, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx = Context(0)
    lib = N.lib()
    for it in range(4):
        t0 = time.perf_counter()
        check(lib.orcd_string_dictionary_decode(ctx.handle, doff.data_ptr(), dbytes.data_ptr(), 1000,
                                                idx.data_ptr(), n, offsets.data_ptr(), out.data_ptr(), total,
                                                N.IN_DEVICE | N.OUT_DEVICE), ctx.handle)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    assert torch.equal(offsets, want_off)
    # spot check of the bytes: a prefix and a suffix against a torch gather
    for lo, hi in ((0, 5000), (n - 5000, n)):
        for i in range(lo, hi, 997):
            s, e = int(want_off[i]), int(want_off[i + 1])
            k = int(idx[i])
            assert torch.equal(out[s:e], dbytes[int(doff[k]):int(doff[k + 1])])
    print("dictionary gather: %d values, %.1f MB out, %.3f ms, %.0f GB/s of output bytes + offsets" %
          (n, total / 1e6, dt * 1e3, (total + 8 * n) / dt / 1e9), flush=True)


if __name__ == "__main__":
    main()
