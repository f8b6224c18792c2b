"""Weight gradient dW = g^T x of the encoder / MLP row layers at their real shapes: one library GEMM against the split-K form
(S row slabs as one batched GEMM + fp32 sum of the partials) used by train_graph._RowsLinear.  Usage: python tools/time_dw_small.py"""
import time

import torch

DEV = 'cuda:0'
SHAPES = [(20000, 256, 256), (25000, 128, 64), (25000, 384, 128), (25000, 1024, 64), (25000, 64, 128), (25000, 128, 256), (25000, 512, 32),
          (25000, 32, 128), (6250, 256, 128), (6250, 2048, 128), (6250, 768, 256), (6250, 1024, 64), (6250, 64, 256), (6250, 256, 512),
          (1560, 512, 256), (1560, 4096, 256), (100000, 512, 32), (100000, 64, 32), (100000, 192, 64)]


def timed(fn, n=20):
    """Device time per call: n calls captured into one HIP graph, the graph replayed (no launch overhead of the host in the figure)."""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(n):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (5 * n) * 1e6


for rows, k, n in SHAPES:
    x = torch.randn(rows, k, device=DEV).to(torch.bfloat16)
    g = torch.randn(rows, n, device=DEV).to(torch.bfloat16)
    line = '{:7d} x[{:4d}] g[{:4d}]: mm {:7.1f} us'.format(rows, k, n, timed(lambda: torch.mm(g.t(), x, out_dtype=torch.float32)))
    for s in (4, 8, 10, 16, 20, 25, 32, 40, 50, 64):
        rs = rows // s
        if rs < 64 or rows % s:
            continue
        main = rs * s

        def split():
            dw = torch.bmm(g[:main].view(s, rs, -1).transpose(1, 2), x[:main].view(s, rs, -1)).sum(0, dtype=torch.float32)
            if main < rows:
                dw = dw + (g[main:].t() @ x[main:]).float()
            return dw
        line += '  S{} {:6.1f}'.format(s, timed(split))
    print(line)
