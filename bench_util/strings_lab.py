"""Experiment: where does the time of the string-bytes pass go?  Times phase 4 of
orcd_string_direct_decode (copy only, validation only, both) against torch's device copy."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench_util import synth  # noqa: E402
from orc_format_amd import Context  # noqa: E402


def main():
    n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000 // 512
    n = n_runs * 512
    lstream, entries, lengths = synth.string_lengths_w6(n_runs, device="cuda")
    total = int(lengths.sum().item())
    del lengths
    data = torch.randint(32, 127, (total,), dtype=torch.uint8, device="cuda")
    out = torch.empty(total, dtype=torch.uint8, device="cuda")
    offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx = Context(0)
    for name, dst, check in (("copy", out.data_ptr(), False), ("check", 0, True), ("copy+check", out.data_ptr(), True)):
        ms = []
        for it in range(6):
            ctx.string_direct_device(lstream.data_ptr(), lstream.numel(), data.data_ptr(), total, n,
                                     offsets.data_ptr(), dst, entries.data_ptr(), entries.numel(), check_utf8=check)
            ms.append([ctx.phase_ms(p) for p in (1, 2, 3, 4)])
        best = min(m[3] for m in ms[1:])
        print("%-11s phases(ms) %s   bytes pass %.3f ms = %.0f GB/s (2 x data)" %
              (name, " ".join("%.3f" % x for x in ms[-1]), best, 2 * total / best / 1e6), flush=True)
    torch.cuda.synchronize()
    for it in range(3):
        t0 = time.perf_counter()
        out.copy_(data)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("torch copy_ %.3f ms = %.0f GB/s (2 x data)" % (dt * 1e3, 2 * total / dt / 1e9))


if __name__ == "__main__":
    main()
