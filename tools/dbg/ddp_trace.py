"""Host / queue times of the staged multi-rank step in a one-rank RCCL group (B = 10): python tools/dbg/ddp_trace.py"""
import json
import os
import sys
import time

os.environ['PPS_SINGLE_RANK_COLLECTIVES'] = '1'
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_workloads as workloads            # noqa: E402
from ppsurf_amd import sharding                # noqa: E402
from ppsurf_amd.fit import HostGcPacer         # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
sharding.init_process_group(dev, backend='nccl')
fit = workloads.FitStepDDP(batch=10, precision='bf16-mixed', device=dev, rank=0)
for _ in range(fit.WARMUP_STEPS):
    fit()
torch.cuda.synchronize()
with HostGcPacer() as pacer:
    t0 = time.perf_counter()
    for _ in range(40):
        fit()
        pacer.tick()
    host = (time.perf_counter() - t0) / 40 * 1e3
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / 40 * 1e3
print('staged step: host issue {:.2f} ms per step, with the final drain {:.2f} ms'.format(host, total))
fit.start_trace()
with HostGcPacer() as pacer:
    for _ in range(40):
        fit()
        pacer.tick()
tr = fit.read_trace()
print(json.dumps({k: tr[k] for k in ('call_ms', 'queue_busy_ms', 'batch_wait_ms', 'loader_wait_ms', 'loader_host_ms')}))
fit.close()
torch.distributed.destroy_process_group()
