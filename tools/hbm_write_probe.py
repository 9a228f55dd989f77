"""What HBM delivers for pure WRITE streams on this box (the training forward and the data-gradient pass are write streams: 5.9 / 5.7 KB
per sample out, next to nothing in): torch fill / copy kernels over buffers far beyond the 256 MiB Infinity Cache, hipEvents.
usage: python tools/hbm_write_probe.py"""
import torch
dev = torch.device("cuda:0")


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for gb in (1, 4, 8):
    n = gb * (1 << 30) // 4
    x = torch.empty(n, device=dev, dtype=torch.float32)
    y = torch.empty(n, device=dev, dtype=torch.float32).normal_()
    ms_fill = timed(lambda: x.fill_(1.5))
    ms_zero = timed(lambda: x.zero_())
    ms_copy = timed(lambda: x.copy_(y))
    ms_read = timed(lambda: y.sum())
    print("%d GiB: fill %.2f TB/s   zero (memset) %.2f TB/s   copy %.2f TB/s (read + write)   sum (pure read) %.2f TB/s" % (
        gb, 4 * n / ms_fill / 1e9, 4 * n / ms_zero / 1e9, 8 * n / ms_copy / 1e9, 4 * n / ms_read / 1e9), flush=True)
    del x, y
