"""Ad-hoc GPU debugging helper (not collected by pytest)."""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import oracle_ctypes  # noqa: E402
import rle_encode as enc  # noqa: E402
from gpu_backend import GpuBackend  # noqa: E402


def nbad(gpu, stream, expect, signed):
    rc, out, _ = gpu.rlev2(stream, expect.size, signed)
    assert rc == 0
    bad = np.nonzero(out != expect)[0]
    return bad, out


def main():
    oracle = oracle_ctypes.load()
    gpu = GpuBackend()
    rng = np.random.default_rng(3)
    stream, expect, starts = enc.random_stream(rng, 50, False, max_len=512)
    bad, _ = nbad(gpu, stream, expect, False)
    print("fresh: bad", bad.size)
    # poison the arena the way the test sequence does
    for seed, n_runs, max_len in [(1, 300, 512), (2, 2000, 12)]:
        r2 = np.random.default_rng(seed)
        s2, e2, st2 = enc.random_stream(r2, n_runs, True, max_len=max_len)
        b2, _ = nbad(gpu, s2, e2, True)
        print("seed", seed, "bad", b2.size)
        bad, _ = nbad(gpu, stream, expect, False)
        print("  then seed-3 stream: bad", bad.size, bad[:6])
    rc, runs_gpu = gpu.runs(stream, expect.size)
    rc, runs_cpu = oracle.runs(stream, expect.size)
    off = 0
    for i, (g, c) in enumerate(zip(runs_gpu, runs_cpu)):
        if i < 8 or g[3] != off:
            print(i, "gpu kind/w/len/out/payload", g, "cpu kind/w/len/byte/bytes", c, "cpu out", off)
        off += c[2]
    # minimal failing prefix
    rc, runs = oracle.runs(stream, expect.size)
    offs = np.cumsum([0] + [r[2] for r in runs])
    for k in range(1, 12):
        end_byte = runs[k][3] if k < len(runs) else len(stream)
        sub = stream[:end_byte]
        bad, out = nbad(gpu, sub, expect[:offs[k]], False)
        print("prefix", k, "runs:", "bad", bad.size, bad[:8])


if __name__ == "__main__":
    main()
