"""Which torch ops the reconstruction DRIVER (region growing, Marching Cubes, clean-up, refinement bookkeeping) spends its device time in, with
the decoder replaced by the analytic occupancy: torch.profiler over one R=257 reconstruction.  Usage: python tools/profile_driver.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ppsurf_amd import reconstruct, synthetic, mcubes
import bench_workloads as workloads

DEV = 'cuda:0'
cloud, norm = synthetic.make_cloud(100000, seed=42, noise=0.0, return_norm=True)
step, bmin_pad, pts_ids = workloads.grid_geometry(cloud, 257)
ids = torch.from_numpy(pts_ids).to(DEV)
field = lambda q: synthetic.bumpy_occupancy(q, norm)


def stages():
    out = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vol = reconstruct.create_volume(field, ids, 257, step, bmin_pad)
    torch.cuda.synchronize(); out['growth'] = time.perf_counter() - t0; t0 = time.perf_counter()
    v, f = mcubes.marching_cubes_torch(vol, 0.0)
    torch.cuda.synchronize(); out['mc'] = time.perf_counter() - t0; t0 = time.perf_counter()
    v = v.to(torch.float32).to(torch.float64)
    v, f = mcubes.clean_mesh_torch(v, f, min_component_faces=6)
    torch.cuda.synchronize(); out['clean1'] = time.perf_counter() - t0; t0 = time.perf_counter()
    v = reconstruct.refine_vertices(field, v, vol, step, bmin_pad, 10)
    torch.cuda.synchronize(); out['refine'] = time.perf_counter() - t0; t0 = time.perf_counter()
    v, f = mcubes.clean_mesh_torch(v, f, min_component_faces=6)
    torch.cuda.synchronize(); out['clean2'] = time.perf_counter() - t0; t0 = time.perf_counter()
    res = v.to(torch.float32).cpu().numpy(), f.cpu().numpy()
    out['download'] = time.perf_counter() - t0
    return out


for _ in range(2):
    stages()
print(' '.join('{} {:.1f} ms'.format(k, v * 1e3) for k, v in stages().items()))
for name, fn in (('growth', lambda: reconstruct.create_volume(field, ids, 257, step, bmin_pad)),):
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        vol = fn()
        torch.cuda.synchronize()
    print(name); print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=14, max_name_column_width=60))
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    v, f = mcubes.marching_cubes_torch(vol, 0.0)
    torch.cuda.synchronize()
print('mc'); print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=12, max_name_column_width=60))
v = v.to(torch.float32).to(torch.float64)
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    mcubes.clean_mesh_torch(v, f, min_component_faces=6)
    torch.cuda.synchronize()
print('clean'); print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=14, max_name_column_width=60))
