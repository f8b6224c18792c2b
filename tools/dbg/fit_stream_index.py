"""Does the fit step's time depend on WHICH pool stream the loader gets?  K dummy torch.cuda.Stream() objects are created (and used once) before FitStep.
    python tools/dbg/fit_stream_index.py K [K ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_workloads as workloads            # noqa: E402
from ppsurf_amd.fit import HostGcPacer         # noqa: E402

k = int(sys.argv[1])
torch.cuda.set_device(0)
x = torch.zeros(1024, device='cuda:0')
dummies = []
for _ in range(k):
    s = torch.cuda.Stream(device='cuda:0')
    with torch.cuda.stream(s):
        x.add_(1.0)
    dummies.append(s)
torch.cuda.synchronize()
fit = workloads.FitStep(batch=10, precision='bf16-mixed', device='cuda:0', graph=True)
for _ in range(8):
    fit()
torch.cuda.synchronize()
with HostGcPacer() as pacer:
    t0 = time.perf_counter()
    for _ in range(40):
        fit()
        pacer.tick()
    torch.cuda.synchronize()
print('K = {:2d} dummy streams: {:.2f} ms per step (loader stream {})'.format(k, (time.perf_counter() - t0) / 40 * 1e3, fit.prefetch.side), flush=True)
fit.close()
