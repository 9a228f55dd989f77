"""Pure HBM write / read / copy rates of this device with plain torch kernels (context for the training kernels' stores)."""
import torch
dev = torch.device("cuda:0")
n = 1 << 30          # 4 GiB of fp32
x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
s = t(lambda: x.fill_(1.0)); print(f"fill   {4 * n / s / 1e12:6.2f} TB/s written")
s = t(lambda: x.zero_()); print(f"zero   {4 * n / s / 1e12:6.2f} TB/s written")
s = t(lambda: y.copy_(x)); print(f"copy   {4 * n / s / 1e12:6.2f} TB/s read + the same written")
s = t(lambda: x.sum()); print(f"sum    {4 * n / s / 1e12:6.2f} TB/s read")
xb = x.view(torch.bfloat16)
s = t(lambda: xb.fill_(1.0)); print(f"fill16 {4 * n / s / 1e12:6.2f} TB/s written")
