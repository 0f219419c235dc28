"""Experiment: does decoding the stripe set as S stripes on T contexts (one HIP stream and one
host thread each) hide the header-scan phases behind the value kernels of other stripes?"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench_util import synth  # noqa: E402
from orc_format_amd import Context  # noqa: E402


def main():
    n_runs = 1_000_000_000 // 512
    run_bytes = 1090
    stream, entries, expect = synth.direct_w17(n_runs, device="cuda")
    out = torch.empty(n_runs * 512, dtype=torch.int64, device="cuda")
    steps = 10
    for n_stripes, n_threads in [(1, 1), (2, 2), (4, 2), (8, 2), (16, 2), (8, 3), (16, 4), (32, 4), (64, 4)]:
        bounds = [(i * n_runs) // n_stripes // 20 * 20 for i in range(n_stripes)] + [n_runs]
        stripes = []
        for i in range(n_stripes):
            r0, r1 = bounds[i], bounds[i + 1]
            ent = (entries[r0 // 20:(r1 + 19) // 20] - r0 * run_bytes).contiguous()
            stripes.append((stream[r0 * run_bytes:r1 * run_bytes], ent, out[r0 * 512:r1 * 512], (r1 - r0) * 512))
        ctxs = [Context(0) for _ in range(n_threads)]

        def work(t, k):
            ctx = ctxs[t]
            for _ in range(k):
                for i in range(t, n_stripes, n_threads):
                    s, e, o, n = stripes[i]
                    ctx.rlev2_device(s.data_ptr(), s.numel(), n, True, o.data_ptr(), e.data_ptr(), e.numel())
            ctx.synchronize()

        def run(k):
            ths = [threading.Thread(target=work, args=(t, k)) for t in range(n_threads)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()

        run(2)
        torch.cuda.synchronize()
        out.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ok = torch.equal(out, expect)
        print("stripes %3d threads %d: %.3f ms/step  %.0f GB/s decoded  %s" %
              (n_stripes, n_threads, dt * 1e3, n_runs * 512 * 8 / dt / 1e9, "ok" if ok else "MISMATCH"), flush=True)
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
