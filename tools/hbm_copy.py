"""Streaming-copy bandwidth of the device (context for the HBM roofline fractions): y = x over 2 GiB, read + write counted."""
import torch

n = 2 * 1024 ** 3 // 8
x = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(20):
    y.copy_(x)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 20
print(f"device copy of {n * 8 / 2 ** 30:.0f} GiB: {ms:.3f} ms -> {2 * n * 8 / ms / 1e6:.0f} GB/s (read + write)")
z = torch.empty(n, dtype=torch.float64, device="cuda")
ev0.record()
for _ in range(20):
    z.fill_(1.0)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 20
print(f"device fill: {n * 8 / ms / 1e6:.0f} GB/s (write only)")
