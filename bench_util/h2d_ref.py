"""Reference point for the file workloads: host-to-device rate of this box out of pinned memory,
one copy at a time and four streams at once (python3 bench_util/h2d_ref.py)."""
import torch, time
n = 1 << 30
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(2): d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("H2D 1 GiB pinned, one stream: %.1f GB/s" % (n / dt / 1e9))
streams = [torch.cuda.Stream() for _ in range(4)]
q = n // 4
t0 = time.perf_counter()
for _ in range(5):
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            d[i * q:(i + 1) * q].copy_(h[i * q:(i + 1) * q], non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("H2D 1 GiB pinned, four streams: %.1f GB/s" % (n / dt / 1e9))
# 4 MiB pieces, as a reader that fills pinned chunks would queue them
p = 4 << 20
t0 = time.perf_counter()
for _ in range(5):
    for o in range(0, n, p):
        d[o:o + p].copy_(h[o:o + p], non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("H2D 1 GiB pinned in 4 MiB pieces, one stream: %.1f GB/s" % (n / dt / 1e9))
