"""Experiment: a stripe of many small columns (64 x 1M rows, Direct w17, device resident) decoded
by T host threads, each with its own context / HIP stream.  Does the per-call cost overlap?"""
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
    n_cols, n_runs = 64, 2000
    n = n_runs * 512
    cols = []
    for c in range(n_cols):
        stream, entries, expect = synth.direct_w17(n_runs, seed=100 + c, device="cuda")
        cols.append((stream, entries, expect, torch.empty(n, dtype=torch.int64, device="cuda")))
    torch.cuda.synchronize()
    for threads in (1, 2, 4, 8, 16):
        ctxs = [Context(0) for _ in range(threads)]

        def work(t):
            ctx = ctxs[t]
            for c in range(t, n_cols, threads):
                s, e, _, o = cols[c]
                ctx.rlev2_device(s.data_ptr(), s.numel(), n, True, o.data_ptr(), e.data_ptr(), e.numel())
            ctx.synchronize()

        best = 1e9
        for rep in range(5):
            ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            best = min(best, time.perf_counter() - t0)
        ok = all(torch.equal(o, x) for _, _, x, o in cols)
        print("threads %2d: %.3f ms per stripe of %d columns, %.0f GB/s decoded  %s" %
              (threads, best * 1e3, n_cols, n_cols * n * 8 / best / 1e9, "ok" if ok else "MISMATCH"), flush=True)
        for c in ctxs:
            c.close()
    # the same stripe as one batch: the columns' streams copied into one buffer, one call
    pieces, slices, pos = [], [], 0
    for s, e, _, _ in cols:
        slices.append((pos, s.numel(), n, e.cpu().numpy().astype("uint64")))
        pieces.append(s)
        pos += s.numel()
    blob = torch.cat(pieces)
    out = torch.empty(n_cols * n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx = Context(0)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        ctx.rlev2_batch((blob.data_ptr(), blob.numel()), slices, True, out_ptr=out.data_ptr(), device=True)
        ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    ok = torch.equal(out, torch.cat([x for _, _, x, _ in cols]))
    print("one batch : %.3f ms per stripe of %d columns, %.0f GB/s decoded  %s" %
          (best * 1e3, n_cols, n_cols * n * 8 / best / 1e9, "ok" if ok else "MISMATCH"), flush=True)


if __name__ == "__main__":
    main()
