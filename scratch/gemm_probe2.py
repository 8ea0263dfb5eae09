import torch
dev = torch.device('cuda')
def bench(fn, name, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:60s} {s.elapsed_time(e)/n*1e3:8.1f} us", flush=True)
M, B = 128064, 16326
for K in (100, 128):
    x = torch.randn(M, K, device=dev); w = torch.randn(256, K, device=dev); dy = torch.randn(M, 256, device=dev)
    bias = torch.randn(256, device=dev)
    bench(lambda: torch.addmm(bias, x, w.t()), f"embed fwd [M,{K}]x[{K},256]")
    bench(lambda: dy.t() @ x, f"embed wgrad dy^T x K={K}")
    k = (M // 8) * 8
    bench(lambda: torch.bmm(dy[:k].view(8, k // 8, 256).transpose(1, 2), x[:k].view(8, k // 8, K)).sum(0), f"embed wgrad 8 slabs K={K}")
    k = (M // 32) * 32
    bench(lambda: torch.bmm(dy[:k].view(32, k // 32, 256).transpose(1, 2), x[:k].view(32, k // 32, K)).sum(0), f"embed wgrad 32 slabs K={K}")
x100 = torch.randn(M, 100, device=dev)
bench(lambda: torch.nn.functional.pad(x100, (0, 28)), "pad 100 -> 128")
dy = torch.randn(M, 256, device=dev)
bench(lambda: dy.sum(0), "bias grad dy.sum(0) [M,256]")
ones = torch.ones(1, M, device=dev)
bench(lambda: ones @ dy, "bias grad ones @ dy [M,256]")
dy2 = torch.randn(B, 47, device=dev); ones2 = torch.ones(1, B, device=dev)
bench(lambda: dy2.sum(0), "bias grad dy.sum(0) [B,47]")
bench(lambda: ones2 @ dy2, "bias grad ones @ dy [B,47]")
print((ones @ dy - dy.sum(0)).abs().max().item())
# wide x0 GEMMs
x0 = torch.randn(B, 256, device=dev); wc = torch.randn(256, 2048, device=dev); d = torch.randn(B, 2048, device=dev)
bench(lambda: x0 @ wc, "x0 @ Wcat [B,256]x[256,2048]")
bench(lambda: d @ wc.t(), "dX0W @ Wcat^T")
bench(lambda: x0.t() @ d, "x0^T @ dX0W")
k = (B // 8) * 8
bench(lambda: torch.bmm(x0[:k].view(8, k // 8, 256).transpose(1, 2), d[:k].view(8, k // 8, 2048)).sum(0), "x0^T @ dX0W 8 slabs")
