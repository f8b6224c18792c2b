"""How fast are the library GEMMs of the fit step's tall-skinny layers?  (dev tool)"""
import torch, sys
dev = 'cuda'
R = 1280000


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for lib in ('default', 'cublas', 'cublaslt'):
    if lib != 'default':
        torch.backends.cuda.preferred_blas_library(lib)
    for dt in (torch.bfloat16, torch.float16):
        x = torch.randn(R, 256, device=dev, dtype=dt); w = torch.randn(256, 256, device=dev, dtype=dt); b = torch.randn(256, device=dev, dtype=dt)
        g = torch.randn(R, 256, device=dev, dtype=dt)
        f1 = t(lambda: torch.nn.functional.linear(x, w, b))
        f2 = t(lambda: x @ w.t())
        f3 = t(lambda: (w @ x.t()))
        f4 = t(lambda: g.t() @ x)            # weight gradient
        f5 = t(lambda: g @ w)                # input gradient
        print('{:9s} {}: linear {:.3f}  x@w.T {:.3f}  w@x.T {:.3f}  dW=g.T@x {:.3f}  dX=g@w {:.3f} ms'.format(lib, str(dt)[6:], f1, f2, f3, f4, f5))

torch.backends.cuda.preferred_blas_library('default')
print('split-K weight gradient  dW = sum_s g[s].T @ x[s]   (bf16 in, fp32 partials)')
for n, k, rows in ((256, 256, 1280000), (64, 256, 1280000), (256, 259, 1280000), (64, 64, 1000000), (128, 64, 1000000), (256, 128, 1000000)):
    g = torch.randn(rows, n, device=dev, dtype=torch.bfloat16); x = torch.randn(rows, k, device=dev, dtype=torch.bfloat16)
    base = t(lambda: g.t() @ x)
    line = 'N={} K={} rows={}: plain {:.3f} ms;'.format(n, k, rows, base)
    for s in (64, 256, 1024):
        rs = rows // s
        f = t(lambda: torch.bmm(g[:rs * s].view(s, rs, n).transpose(1, 2), x[:rs * s].view(s, rs, k)).float().sum(0))
        f32 = t(lambda: torch.baddbmm(torch.zeros(s, n, k, device=dev), g[:rs * s].view(s, rs, n).transpose(1, 2).float(), x[:rs * s].view(s, rs, k).float()).sum(0)) if s == 256 and n * k <= 65536 and False else 0
        line += '  S={} {:.3f}'.format(s, f)
    print(line)
