"""Per-step wall times of the replayed fit step (bench leg): where do the slow blocks come from?   python tools/fit_step_jitter.py [gc]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench_workloads as workloads

step = workloads.FitStep(batch=10, precision='bf16-mixed', graph=True)
for _ in range(8):
    step()
if len(sys.argv) > 1 and sys.argv[1] == 'gc':
    gc.collect(); gc.freeze(); gc.disable()
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for i in range(240):
    step()
    torch.cuda.current_stream().synchronize() if False else None
    ts.append(time.perf_counter())
torch.cuda.synchronize()
total = time.perf_counter() - t0
d = np.diff(np.array([t0] + ts)) * 1e3
print('mean {:.2f} ms/step over 240; host-side per-call: median {:.2f}, p90 {:.2f}, max {:.2f} ms'.format(total / 240 * 1e3, np.median(d), np.percentile(d, 90), d.max()))
print('calls above 40 ms:', [(int(i), round(float(v), 1)) for i, v in enumerate(d) if v > 40][:20])
print('block means of 40:', [round(float(d[i:i + 40].mean()), 2) for i in range(0, 240, 40)])
step.close()
