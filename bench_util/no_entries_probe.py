"""Times decodes of streams that come without entry points (GPU box; prints milliseconds)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from bench_util import synth  # noqa: E402
from orc_format_amd import Context  # noqa: E402


def timed(fn, sync, reps=3):
    fn()
    sync()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        sync()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def main():
    ctx = Context(0)

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
    stream, entries, values, _ = synth.mixed_short_runs(1_000_000, device="cuda")
    n = values.numel()
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    for name, e in (("with entries", entries), ("no entries", None)):
        ms = timed(lambda: ctx.rlev2_device(stream.data_ptr(), stream.numel(), n, False, out.data_ptr(),
                                            e.data_ptr() if e is not None else 0, e.numel() if e is not None else 0), sync)
        assert torch.equal(out, values)
        print("mixed_short_runs 1e6 runs, %s: %.3f ms" % (name, ms), flush=True)
    del stream, entries, values, out
    n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000_000 // 8 // 128
    stream, entries, packed = synth.byte_rle_mix(n_blocks, device="cuda")
    bits = packed.numel() * 8
    outb = torch.empty(bits, dtype=torch.uint8, device="cuda")
    for name, e in (("with entries", entries), ("no entries", None)):
        ms = timed(lambda: ctx.boolean_device(stream.data_ptr(), stream.numel(), bits, outb.data_ptr(), 0,
                                              e.data_ptr() if e is not None else 0, e.numel() if e is not None else 0), sync)
        print("boolean %d bits, %s: %.3f ms" % (bits, name, ms), flush=True)


main()
