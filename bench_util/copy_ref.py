"""Reference point for the string-bytes kernel: what the runtime's own device-to-device copy of
8 GB achieves on the same box (python3 bench_util/copy_ref.py)."""
import torch, time
n = 8_000_000_000
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
a.fill_(65)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): b.copy_(a)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print("D2D copy of 8 GB: %.3f ms, %.2f TB/s read+write" % (ms, 2 * n / ms / 1e9))
a64 = a.view(torch.int64); b64 = b.view(torch.int64)
s.record()
for _ in range(10): torch.add(a64, 1, out=b64)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print("elementwise add int64 8 GB: %.3f ms, %.2f TB/s" % (ms, 2 * n / ms / 1e9))
