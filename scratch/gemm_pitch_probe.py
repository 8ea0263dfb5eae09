"""dH = dY @ W^T of a GCNII layer (16k x 256 x 256) with dY at the pitch of its own matrix (288
floats) against dY as a column block of the side-by-side gradient stack (pitch 2080), cache-warm
and with the caches flushed between calls (as inside a training step)."""
import torch

dev = torch.device("cuda")
B, C, L = 16326, 256, 8
w = torch.randn(C, C, device=dev)
out = torch.empty(B, 288, device=dev)[:, :C]
flush = torch.empty(1 << 28, device=dev)          # 1 GiB of floats


def timeit(fn, cold, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        if cold:
            flush.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for pitch in (256, 288, 2080, 2048 + 288):
    stack = torch.randn(B, pitch, device=dev)
    dy = stack[:, :C] if pitch < 2000 else stack[:, 3 * C:4 * C]
    h = torch.randn(B, 288, device=dev)[:, :C]
    for cold in (False, True):
        t_dh = timeit(lambda: torch.mm(dy, w.t(), out=out), cold)
        t_fw = timeit(lambda: dy.addmm_(h, w), cold)
        k = (B // 8) * 8
        t_dw = timeit(lambda: torch.bmm(h[:k].unflatten(0, (8, k // 8)).transpose(1, 2),
                                        dy[:k].unflatten(0, (8, k // 8))), cold)
        print(f"pitch {pitch:5d} {'cold' if cold else 'warm'}: dH {t_dh:6.1f} us   term += h W {t_fw:6.1f} us   "
              f"dW slabs {t_dw:6.1f} us", flush=True)
