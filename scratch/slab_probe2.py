"""a^T @ b over row slabs when b (or a) is a column block of a wider matrix (row pitch of the
whole matrix): does hipBLASLt keep its fast kernel?"""
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
    return torch.bmm(a[:k].unflatten(0, (slabs, k // slabs)).transpose(1, 2),
                     b[:k].unflatten(0, (slabs, k // slabs)))
rows, m, n = 16326, 256, 256
a = torch.randn(rows, m, device=dev)
for pitch in (256, 288, 2048, 2080, 2112, 2304):
    wide = torch.randn(rows, pitch, device=dev)
    b = wide[:, :n]
    res = {}
    for s in (6, 8, 9):
        res[s] = round(timeit(lambda: slab(a, b, s)), 1)
    res["b^T a (swapped) 9"] = round(timeit(lambda: slab(b, a, 9)), 1)
    res["dh = b @ W"] = round(timeit(lambda: b @ a[:256]), 1)
    print("pitch of b", pitch, res, flush=True)
