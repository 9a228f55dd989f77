"""How should the S-deep weight-gradient GEMM be fed to the library? (diagnostic)"""
import time, torch
dev = torch.device("cuda:0")
S, M, N = 4096 * 192, 256, 256
dy = torch.randn(S, M, device=dev, dtype=torch.bfloat16)
x = torch.randn(S, N, device=dev, dtype=torch.bfloat16)
def T(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
ref = None
def report(name, fn):
    global ref
    t, r = T(fn)
    r = r.float()
    if ref is None: ref = r
    print(f"{name:44s} {t:8.3f} ms   {2*S*M*N/t/1e9:7.1f} TFLOP/s   rel err {((r-ref).norm()/ref.norm()).item():.2e}")
report("mm(dy.t(), x) fp32 out", lambda: torch.mm(dy.t(), x, out_dtype=torch.float32))
report("mm(dy.t(), x) bf16 out", lambda: torch.mm(dy.t(), x))
report("mm(dy.t().contiguous(), x)", lambda: torch.mm(dy.t().contiguous(), x, out_dtype=torch.float32))
for nb in (16, 64, 256, 1024):
    report(f"bmm over {nb} S-slabs, fp32 sum", lambda nb=nb: torch.bmm(dy.view(nb, S // nb, M).transpose(1, 2), x.view(nb, S // nb, N)).float().sum(0))
    try:
        report(f"bmm over {nb} S-slabs (fp32 out), sum", lambda nb=nb: torch.bmm(dy.view(nb, S // nb, M).transpose(1, 2), x.view(nb, S // nb, N), out_dtype=torch.float32).sum(0))
    except Exception as e:
        print("bmm out_dtype unsupported:", type(e).__name__)
