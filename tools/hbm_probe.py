"""Practical HBM read / copy rate of the box (the ceiling the streaming kernels are measured against)."""
import time, torch
x = torch.empty(1 << 30, device="cuda", dtype=torch.float32).normal_()  # 4 GiB
y = torch.empty_like(x)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
r = t(lambda: x.sum())
c = t(lambda: y.copy_(x))
print(f"read (sum of 4 GiB): {x.numel() * 4 / r / 1e12:.2f} TB/s; copy: {2 * x.numel() * 4 / c / 1e12:.2f} TB/s (read + write)")
w = t(lambda: y.fill_(1.5))
z = t(lambda: y.zero_())
print(f"write only (fill of 4 GiB): {y.numel() * 4 / w / 1e12:.2f} TB/s; zero_: {y.numel() * 4 / z / 1e12:.2f} TB/s")
