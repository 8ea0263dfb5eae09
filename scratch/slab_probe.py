"""a^T @ b for tall operands as a batched product over row slabs: slab counts and remainders."""
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
def slab(a, b, slabs):
    rows = a.size(0); k = rows // slabs * slabs
    out = torch.bmm(a[:k].unflatten(0, (slabs, k // slabs)).transpose(1, 2),
                    b[:k].unflatten(0, (slabs, k // slabs))).sum(0)
    if k < rows: out.addmm_(a[k:].t(), b[k:])
    return out
for rows, m, n in ((16326, 256, 256), (16327, 256, 256), (16326, 256, 2048), (128064, 256, 100), (125690, 256, 100), (16326, 47, 256)):
    a = torch.randn(rows, m, device=dev); b = torch.randn(rows, n, device=dev)
    res = {"one": round(timeit(lambda: a.t() @ b), 1)}
    for s in (6, 8, 9, 16, 18, 29, 32, 48, 64):
        res[s] = (round(timeit(lambda: slab(a, b, s)), 1), rows % s)
    print(rows, m, n, res, flush=True)
