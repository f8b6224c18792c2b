import sys, time
sys.path.insert(0,'.')
import numpy as np, torch
from ppsurf_amd import ops
import bench_workloads as workloads
from ppsurf_amd.synthetic import make_cloud
DEV='cuda:0'
base = make_cloud(100000, seed=42)
chunks,_ = workloads.band_chunks(base, 257, 50000, DEV)
q = chunks[10]
for n in (25000, 50000, 100000, 200000, 400000):
    cloud = make_cloud(n, seed=42)
    pts = torch.from_numpy(cloud).to(DEV)
    kb = ops.KnnBlocks(pts)
    for k in (16, 64):
        for _ in range(3): kb.query(q, k)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(20): kb.query(q, k)
        torch.cuda.synchronize(); print('N={:7d} nb={:5d} k={:3d}: {:.3f} ms'.format(n, kb.nb, k, (time.perf_counter()-t0)/20*1e3))
