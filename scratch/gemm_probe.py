import torch, time
dev = torch.device('cuda')
B, F = 16326, 256
a = torch.randn(B, F, device=dev); b = torch.randn(B, F, device=dev)
def bench(fn, name, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:40s} {s.elapsed_time(e)/n*1e3:8.1f} us")
print('preferred', torch.backends.cuda.preferred_blas_library())
bench(lambda: a.t() @ b, 'a.T@b default')
bench(lambda: a @ b[:256].t(), 'a@W default (fwd shape)')
for S in (4, 8, 16, 32):
    k = (B // S) * S
    def f(S=S, k=k):
        r = torch.bmm(a[:k].view(S, k//S, F).transpose(1,2), b[:k].view(S, k//S, F)).sum(0)
        if k < B: r = r.addmm_(a[k:].t(), b[k:])
        return r
    bench(f, f'bmm split-K S={S}')
try:
    torch.backends.cuda.preferred_blas_library('cublas')
    print('preferred', torch.backends.cuda.preferred_blas_library())
    bench(lambda: a.t() @ b, 'a.T@b rocblas')
    bench(lambda: a @ b[:256].t(), 'a@W rocblas (fwd shape)')
except Exception as ex:
    print('rocblas fail', ex)
ref = (a.double().t() @ b.double())
torch.backends.cuda.preferred_blas_library('cublaslt')
k = (B//8)*8
r = torch.bmm(a[:k].view(8, k//8, F).transpose(1,2), b[:k].view(8, k//8, F)).sum(0).addmm_(a[k:].t(), b[k:])
print('err bmm', (r.double()-ref).abs().max().item(), 'err mm', ((a.t()@b).double()-ref).abs().max().item())
