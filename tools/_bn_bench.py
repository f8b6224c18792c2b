import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppsurf_amd import train_ops, _lib
DEV = 'cuda:0'
def timed(fn, n=20):
    if os.environ.get('PROF'):
        for _ in range(30): fn()
        torch.cuda.synchronize(); return 0.0
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(n): fn()
    graph.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (5 * n) * 1e6
for rows, c in [(100000, 64), (25000, 128), (6250, 256), (1560, 512), (20000, 256)]:
    x = torch.randn(rows, c, device=DEV).bfloat16()
    w, b = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    t1 = timed(lambda: train_ops.bn_act(x, w, b, rm, rv, 0.1, 1e-5, True))
    t2 = timed(lambda: train_ops.bn_act(x, w, b, None, None, 0.1, 1e-5, True))
    print(rows, c, 'with running stats {:.1f} us, without {:.1f} us'.format(t1, t2))
