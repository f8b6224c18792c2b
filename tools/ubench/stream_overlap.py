"""Can two independent kernel chains overlap on this GPU / runtime?  Chain A: many short low-occupancy kernels (the encoder of the fit step looks like
this: ~1000 launches of 5-20 us), chain B: a few bandwidth-bound kernels (the decoder's row layers).  Timed: one stream eager, two streams eager, ONE
captured graph with a forked side stream, TWO captured graphs replayed on two streams.
    python tools/ubench/stream_overlap.py"""
import time
import torch

dev = torch.device('cuda')
small = [torch.randn(4096, 64, device=dev) for _ in range(4)]
w = torch.randn(64, 64, device=dev)
big = torch.randn(64 * 1024 * 1024, device=dev)          # 256 MB
big2 = torch.empty_like(big)


def chain_a(n=300):
    x = small[0]
    for i in range(n):
        x = torch.relu(x @ w) * 0.5 + small[i & 3]
    return x


def chain_b(n=12):
    for _ in range(n):
        torch.mul(big, 1.0001, out=big2)
    return big2


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


side = torch.cuda.Stream()


def serial():
    chain_a(); chain_b()


def two_streams():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        chain_b()
    chain_a()
    cur.wait_stream(side)


print('eager  A alone %.2f ms, B alone %.2f ms, serial %.2f ms, two streams %.2f ms' % (timeit(chain_a), timeit(chain_b), timeit(serial), timeit(two_streams)))

# graphs
def capture(fn):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()                                    # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    return g

ga, gb, gs, gf = capture(chain_a), capture(chain_b), capture(serial), capture(two_streams)
print('graph  A alone %.2f ms, B alone %.2f ms, serial graph %.2f ms, forked graph %.2f ms' % (timeit(ga.replay), timeit(gb.replay), timeit(gs.replay), timeit(gf.replay)))

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two_graphs():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        ga.replay()
    with torch.cuda.stream(s2):
        gb.replay()
    cur.wait_stream(s1); cur.wait_stream(s2)


print('two graphs on two streams %.2f ms' % timeit(two_graphs))
