"""d x_0 = dY_cat [B, 2048] @ W2cat^T [2048, 256] over the batch sizes of the headline run: which
row counts get hipBLASLt's faster kernel?  Flushed caches between calls."""
import torch
dev = torch.device("cuda")
w = torch.randn(256, 2048, device=dev) * 0.05
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def t(fn, n=12):
    for _ in range(3):
        fn()
    s = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for i in range(n):
        flush.zero_()
        s[i].record(); fn(); e[i].record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in zip(s, e))[n // 2] * 1e3


for b in (16128, 16256, 16300, 16320, 16326, 16327, 16330, 16340, 16384, 16400, 16512, 16640):
    buf = torch.randn(b, 2080, device=dev)
    d = buf[:, :2048]
    dc = d.contiguous()
    out = torch.empty(b, 256, device=dev)
    print(f"B {b}: stack (pitch 2080) {t(lambda: torch.mm(d, w.t(), out=out)):6.1f} us | contiguous "
          f"{t(lambda: torch.mm(dc, w.t(), out=out)):6.1f} us | w^T stored [2048,256] "
          f"{t(lambda wt=w.t().contiguous(): torch.mm(d, wt, out=out)):6.1f} us", flush=True)
