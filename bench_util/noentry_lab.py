"""Experiment: the headline column decoded without the entry points of a ROW_INDEX."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench_util import synth
from orc_format_amd import Context

n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000 // 512
stream, entries, expect = synth.direct_w17(n_runs, device="cuda")
out = torch.empty(n_runs * 512, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
ctx = Context(0)
for use in (True, False):
    best = 1e9
    for it in range(3):
        t0 = time.perf_counter()
        ctx.rlev2_device(stream.data_ptr(), stream.numel(), n_runs * 512, True, out.data_ptr(),
                         entries.data_ptr() if use else 0, entries.numel() if use else 0)
        ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    assert torch.equal(out, expect)
    print("%s entry points: %.3f ms, %.0f GB/s decoded" % ("with" if use else "without", best * 1e3, n_runs * 4096 / best / 1e9), flush=True)
