"""Device kernels of ONE batch preparation of the fit step (patch search, support sampling, the 13 id tables per cloud, projection table, CSR sorts):
torch.profiler over workloads.FitStep._prepare, every kernel above 20 us listed in launch order.  Usage: python tools/prep_kernels.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_workloads as workloads          # noqa: E402

step = workloads.FitStep(overlap_prep=False)
for _ in range(3):
    step._prepare(0)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity          # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step._prepare(1)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type.name != 'CPU']
print('device total {:.2f} ms in {} kernels'.format(sum(e.device_time for e in evs) / 1e3, len(evs)))
for e in evs:
    if e.device_time > 20:
        print('{:8.1f} us  {}'.format(e.device_time, e.name[:100]))
t0 = time.perf_counter()
for _ in range(10):
    step._prepare(0)
torch.cuda.synchronize()
print('wall {:.2f} ms per batch (one stream, nothing beside it)'.format((time.perf_counter() - t0) * 100))
