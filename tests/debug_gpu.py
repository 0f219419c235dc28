"""Ad-hoc GPU debugging helper (not collected by pytest): decodes random streams restricted to
subsets of run kinds and prints where the HIP decoder first departs from the oracle."""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import oracle_ctypes  # noqa: E402
import rle_encode as enc  # noqa: E402
from gpu_backend import GpuBackend  # noqa: E402


def main():
    oracle = oracle_ctypes.load()
    gpu = GpuBackend()
    cases = [((1,), 512), ((3,), 512), ((2,), 512), ((0,), 512), ((1, 3), 512), ((1, 2), 512),
             ((0, 1, 2, 3), 512), ((0, 1, 2, 3), 40)]
    for kinds, max_len in cases:
        for seed in (1, 2, 3, 4):
            for signed in (False, True):
                rng = np.random.default_rng(seed)
                stream, expect, starts = enc.random_stream(rng, 300, signed, kinds=kinds, max_len=max_len)
                rc, out, _ = gpu.rlev2(stream, expect.size, signed)
                if rc != 0:
                    print("kinds", kinds, "seed", seed, "signed", signed, "rc", rc)
                    continue
                bad = np.nonzero(out != expect)[0]
                if bad.size == 0:
                    continue
                rc, runs = oracle.runs(stream, expect.size)
                offs = np.cumsum([0] + [r[2] for r in runs])
                print("kinds", kinds, "max_len", max_len, "seed", seed, "signed", signed,
                      "bad", bad.size, "of", expect.size)
                brk = np.nonzero(np.diff(bad) > 1)[0]
                for r in np.split(bad, brk + 1)[:6]:
                    b = int(r[0])
                    ri = int(np.searchsorted(offs, b, side="right") - 1)
                    k, w, l, bo, bl = runs[ri]
                    print("   range", b, "..", int(r[-1]), "run", ri, "kind", k, "w", w, "len", l,
                          "out_off", int(offs[ri]), "pos", int(b - offs[ri]), "got",
                          hex(int(out[b]) & (2**64 - 1)), "exp", hex(int(expect[b]) & (2**64 - 1)))


if __name__ == "__main__":
    main()
