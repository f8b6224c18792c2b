"""Is the latent loop (100 encoder passes of a 100k-point cloud, 10 per batch) host- or device-bound?  Wall time vs the sum of kernel times and
the launch count (torch.profiler).  Usage: python tools/profile_latent_loop.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench_workloads as workloads
from ppsurf_amd.synthetic import make_cloud

DEV = 'cuda:0'
model = workloads.make_model(device=DEV)
pts = torch.from_numpy(make_cloud(100000, seed=42)).to(DEV).t().contiguous()
for _ in range(2):
    model.encode_latents(pts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    model.encode_latents(pts)
torch.cuda.synchronize()
print('latent loop: {:.1f} ms wall per shape'.format((time.perf_counter() - t0) / 3 * 1e3))
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    model.encode_latents(pts)
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
print('device kernels: {} launches, {:.1f} ms summed'.format(len(ev), sum(e.device_time for e in ev) / 1e3))
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=25, max_name_column_width=70))
