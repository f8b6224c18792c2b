"""pps_patch_attn_fwd / _bwd alone at the fit step's shape (20 000 patches x 50 points x 256 channels, bf16): device time by HIP-graph replay.
PPS_LIB_VARIANT selects an ablation build.  Usage: python tools/time_patch_attn.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppsurf_amd import _lib          # noqa: E402

DEV = 'cuda:0'
L = _lib.lib()
Q, K, C = 20000, 50, 256
h = torch.randn(Q, K, C, device=DEV).to(torch.bfloat16)
v = torch.randn(C, device=DEV) * 0.1
dp = torch.randn(Q, C, device=DEV)
pooled = torch.empty(Q, C, device=DEV)
dh = torch.empty_like(h)
dv_part = torch.empty(L.pps_patch_attn_partials(Q), C, device=DEV)


def timed(fn, n=20):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn(side.cuda_stream)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(n):
            fn(torch.cuda.current_stream().cuda_stream)
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (5 * n) * 1e3


fwd = timed(lambda st: _lib.check(L.pps_patch_attn_fwd(h.data_ptr(), v.data_ptr(), Q, K, C, 1, pooled.data_ptr(), st), 'fwd'))
bwd = timed(lambda st: _lib.check(L.pps_patch_attn_bwd(h.data_ptr(), v.data_ptr(), dp.data_ptr(), Q, K, C, 1, dh.data_ptr(), dv_part.data_ptr(), st), 'bwd'))
gb = Q * K * C * 2 / 1e9
print('patch_attn fwd {:.3f} ms ({:.2f} TB/s of h), bwd {:.3f} ms ({:.2f} TB/s of h + dh)'.format(fwd, gb / fwd, bwd, 2 * gb / bwd))

# attention pooling of the interpolation head: 20 000 queries x 64 neighbours, 64 heads, 256 channels, h stored before its ReLU
K2, H = 64, 64
qy = torch.randn(Q, K2, H, device=DEV).to(torch.bfloat16)
h2 = torch.randn(Q, K2, C, device=DEV).to(torch.bfloat16)
dp2 = torch.randn(Q, C, device=DEV).to(torch.bfloat16)
pooled2 = torch.empty(Q, C, device=DEV, dtype=torch.bfloat16)
dqy, dh2 = torch.empty_like(qy), torch.empty_like(h2)
fwd = timed(lambda st: _lib.check(L.pps_attn_pool_fwd(qy.data_ptr(), h2.data_ptr(), Q, K2, H, C, 1, 1, pooled2.data_ptr(), st), 'fwd'))
bwd = timed(lambda st: _lib.check(L.pps_attn_pool_bwd(qy.data_ptr(), h2.data_ptr(), dp2.data_ptr(), Q, K2, H, C, 1, 1, dqy.data_ptr(), dh2.data_ptr(), st), 'bwd'))
gf = Q * K2 * (C + H) * 2 / 1e9
print('attn_pool fwd {:.3f} ms ({:.2f} TB/s of qy + h), bwd {:.3f} ms ({:.2f} TB/s of qy + h + dqy + dh)'.format(fwd, gf / fwd, bwd, 2 * gf / bwd))
