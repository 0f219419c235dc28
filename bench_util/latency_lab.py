"""Experiment: cost of one decode call on small device-resident streams (call overhead)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench_util import synth  # noqa: E402
from orc_format_amd import Context  # noqa: E402


def main():
    ctx = Context(0)
    for n_runs in (1, 20, 200, 2000, 20000):
        stream, entries, expect = synth.direct_w17(n_runs, device="cuda")
        n = n_runs * 512
        out = torch.empty(n, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for reps in (3, 200):
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.rlev2_device(stream.data_ptr(), stream.numel(), n, True, out.data_ptr(),
                                 entries.data_ptr(), entries.numel())
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / reps
        assert torch.equal(out, expect)
        print("%8d values: %.1f us per call, %.2f GB/s decoded" % (n, dt * 1e6, n * 8 / dt / 1e9), flush=True)


if __name__ == "__main__":
    main()
