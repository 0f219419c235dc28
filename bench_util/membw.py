"""Memory-bandwidth yardsticks on the current GPU (torch ops): fill, copy, read-reduce."""
import torch


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    n = 1_000_000_000
    a = torch.empty(n, dtype=torch.int64, device="cuda")
    b = torch.empty(n // 2, dtype=torch.int64, device="cuda")
    c = torch.empty(n // 2, dtype=torch.int64, device="cuda")
    ms = timeit(lambda: a.fill_(7))
    print("fill 8 GB: %.3f ms -> %.0f GB/s written" % (ms, 8e9 / ms / 1e6))
    ms = timeit(lambda: a.zero_())
    print("zero 8 GB: %.3f ms -> %.0f GB/s written" % (ms, 8e9 / ms / 1e6))
    ms = timeit(lambda: c.copy_(b))
    print("copy 4 GB -> 4 GB: %.3f ms -> %.0f GB/s read+written" % (ms, 8e9 / ms / 1e6))
    ms = timeit(lambda: a.sum())
    print("sum 8 GB: %.3f ms -> %.0f GB/s read" % (ms, 8e9 / ms / 1e6))
    ms = timeit(lambda: torch.add(b, 1, out=c))
    print("add 4 GB -> 4 GB: %.3f ms -> %.0f GB/s read+written" % (ms, 8e9 / ms / 1e6))


if __name__ == "__main__":
    main()
