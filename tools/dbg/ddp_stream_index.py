"""The multi-rank fit step (fit.StagedStep on a pool stream of its own + the loader's side stream) in a ONE-rank RCCL group: does its time depend on
how many streams were made before it (K1) or between its stream and the loader's (K2)?   python tools/dbg/ddp_stream_index.py K1 K2"""
import os
import sys
import time

os.environ['PPS_SINGLE_RANK_COLLECTIVES'] = '1'
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_workloads as workloads            # noqa: E402
from ppsurf_amd import sharding, data          # noqa: E402
from ppsurf_amd.fit import HostGcPacer         # noqa: E402

k1, k2 = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
sharding.init_process_group(dev, backend='nccl')
x = torch.zeros(1024, device=dev)
keep = []


def dummies(k):
    for _ in range(k):
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            x.add_(1.0)
        keep.append(s)
    torch.cuda.synchronize()


dummies(k1)
orig = data._loader_stream


def patched(device):
    dummies(k2)                                  # streams made between StagedStep's own stream and the loader's
    return orig(device)


data._loader_stream = patched
fit = workloads.FitStepDDP(batch=10, precision='bf16-mixed', device=dev, rank=0)
for _ in range(fit.WARMUP_STEPS):
    fit()
torch.cuda.synchronize()
with HostGcPacer() as pacer:
    t0 = time.perf_counter()
    for _ in range(40):
        fit()
        pacer.tick()
    torch.cuda.synchronize()
print('K1 = {} K2 = {}: {:.2f} ms per staged step (B = 10, one rank, RCCL)'.format(k1, k2, (time.perf_counter() - t0) / 40 * 1e3), flush=True)
fit.close()
torch.distributed.destroy_process_group()
