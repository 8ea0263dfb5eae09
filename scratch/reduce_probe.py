"""Column sums of tall matrices (bias gradients): torch's reduction against GEMV / GEMM forms."""
import torch
dev = torch.device("cuda")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for rows, cols in ((16326, 47), (16326, 256), (128064, 256)):
    dy = torch.randn(rows, cols, device=dev)
    ones = torch.ones(rows, device=dev)
    ones_row = torch.ones(1, rows, device=dev)
    ref = dy.sum(0)
    forms = {"sum(0)": lambda: dy.sum(0), "mv(dy.t, 1)": lambda: torch.mv(dy.t(), ones),
             "1 @ dy": lambda: ones_row @ dy,
             "two-stage": lambda: dy[: rows // 64 * 64].view(64, rows // 64, cols).sum(1).sum(0)}
    print(rows, cols, {k: round(timeit(f), 1) for k, f in forms.items()},
          "err", float((torch.mv(dy.t(), ones) - ref).abs().max()), flush=True)
