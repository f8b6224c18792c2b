"""Latent loop of one 100k-point shape: reference random stream (CPU permutations) vs device permutations; phase split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppsurf_amd import spatial
import bench_workloads as workloads
from ppsurf_amd.synthetic import make_cloud

DEV = 'cuda:0'
model = workloads.make_model(device=DEV)
pts = torch.from_numpy(make_cloud(100000, seed=1)).to(DEV).t().contiguous()
for rng in ('reference', 'device', 'reference', 'device'):
    model.latent_rng = rng
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.encode_latents(pts)
    torch.cuda.synchronize(); print(rng, 'total {:.3f} s'.format(time.perf_counter() - t0))
# phase split of one batch of 10 subsets
subs = [torch.randperm(100000, device=DEV)[:10000] for _ in range(10)]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data = {'pts': torch.stack([pts[:, i] for i in subs])}
    torch.cuda.synchronize(); t1 = time.perf_counter()
    data.update(spatial.get_fkaconv_ids(data))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    lat = model.network.encoder.forward_batch_point_major(data)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print('gather {:.1f} ms, ids {:.1f} ms, encoder {:.1f} ms'.format((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
