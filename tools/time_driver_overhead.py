"""Driver-only time of one R=257 reconstruction: region growing, Marching Cubes, clean-up, refinement with the decoder replaced by the
analytic occupancy (no kernels of the network run) -- what the torch-op driver itself costs per shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppsurf_amd import reconstruct, synthetic, mcubes
import bench_workloads as workloads

DEV = 'cuda:0'
cloud, norm = synthetic.make_cloud(100000, seed=42, noise=0.0, return_norm=True)
step, bmin_pad, pts_ids = workloads.grid_geometry(cloud, 257)
ids = torch.from_numpy(pts_ids).to(DEV)
n = [0]
def field(q):
    n[0] += q.shape[0]
    return synthetic.bumpy_occupancy(q, norm)
for rep in range(3):
    n[0] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vol = reconstruct.create_volume(field, ids, 257, step, bmin_pad)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    v, f = mcubes.marching_cubes_torch(vol, 0.0)
    v = v.to(torch.float32).to(torch.float64)
    v, f = mcubes.clean_mesh_torch(v, f, min_component_faces=6)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    v = reconstruct.refine_vertices(field, v, vol, step, bmin_pad, 10)
    v, f = mcubes.clean_mesh_torch(v, f, min_component_faces=6)
    out = v.to(torch.float32).cpu().numpy(), f.cpu().numpy()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print('growth {:.1f} ms ({} queries), MC + clean {:.1f} ms, refine + clean + download {:.1f} ms ({} vertices)'.format(
        (t1 - t0) * 1e3, n[0], (t2 - t1) * 1e3, (t3 - t2) * 1e3, v.shape[0]))
