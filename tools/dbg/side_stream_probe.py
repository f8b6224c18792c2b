"""Which pool streams disturb the current stream least?  For 8 consecutive torch.cuda.Stream() objects: a burst of small dependent kernels on the current
stream, timed alone and beside a train of big kernels on the candidate; then the config-3 FitStep with that candidate as its loader stream.
    python tools/dbg/side_stream_probe.py [n_candidates] [--fit]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
small = torch.zeros(1 << 18, device=dev)                 # 1 MB
big = torch.zeros(1 << 26, device=dev)                   # 256 MB


def burst(k=300):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k):
        small.mul_(1.0001)
    b.record()
    return a, b


def probe(cand):
    main = torch.cuda.current_stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(cand):
        for _ in range(24):
            big.mul_(1.0001)
    a, b = burst()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


for _ in range(3):
    a, b = burst()
torch.cuda.synchronize()
a, b = burst()
torch.cuda.synchronize()
print('solo burst: {:.3f} ms'.format(a.elapsed_time(b)), flush=True)
cands = [torch.cuda.Stream(device=dev, priority=0) for _ in range(n)]
res = []
for i, c in enumerate(cands):
    t = [probe(c) for _ in range(3)]
    res.append(min(t))
    print('candidate {}: burst beside it {:.3f} / {:.3f} / {:.3f} ms'.format(i, *t), flush=True)
if '--fit' in sys.argv:
    import bench_workloads as workloads
    from ppsurf_amd.fit import HostGcPacer
    for i in (int(x) for x in os.environ.get('FIT_CANDS', '0,1,2,3,4,5,6,7').split(',')):
        fit = workloads.FitStep(batch=10, precision='bf16-mixed', device='cuda:0', graph=True)
        fit.prefetch.side = cands[i]
        for _ in range(8):
            fit()
        torch.cuda.synchronize()
        with HostGcPacer() as pacer:
            t0 = time.perf_counter()
            for _ in range(40):
                fit()
                pacer.tick()
            torch.cuda.synchronize()
        print('FitStep with candidate {} as the loader stream: {:.2f} ms per step (probe {:.3f} ms)'.format(i, (time.perf_counter() - t0) / 40 * 1e3, res[i]), flush=True)
        fit.close()
        del fit
