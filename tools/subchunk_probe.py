import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench_workloads as workloads
from ppsurf_amd.decoder import DecoderPlan, ChunkPipeline
from ppsurf_amd.synthetic import make_cloud, make_latents, network_state_dict
DEV='cuda:0'
sd = network_state_dict('ppsurf')
cloud = make_cloud(100000, seed=42)
pts = torch.from_numpy(cloud).to(DEV)
lat = torch.from_numpy(make_latents(256, 100000, seed=77)[0]).to(DEV)
chunks, _ = workloads.band_chunks(cloud, 257, 400000, DEV)
names = ('interp_pool', 'pointnet_stn_rows', 'pointnet_stn_fc', 'pointnet_feat_rows', 'decode_tail')
for dtype in ('f32','f16x3'):
  plan = DecoderPlan(sd, DEV, dtype=dtype)
  table = plan.point_table(lat)
  for sub in (50000, 100000, 200000, 400000):
    pipe = ChunkPipeline(plan, table, pts, pts, 64, 50, same_cloud=True, max_chunk=sub)
    n = 1
    reps = 10
    ev = [[workloads.HipEvents(6) for _ in range(n)] for _ in range(reps)]
    for c in chunks[:1]:
        pipe.run([c[:sub].contiguous()])
    torch.cuda.synchronize()
    for i in range(reps):
        c = chunks[i % len(chunks)]
        pipe.run([c[j*sub:(j+1)*sub].contiguous() for j in range(n)], stage_events=[e.arr for e in ev[i]])
    torch.cuda.synchronize()
    ms = {nm: float(np.median([sum(e.elapsed_ms(j, j + 1) for e in ev[i]) for i in range(reps)])) for j, nm in enumerate(names)}
    print(dtype, sub, ' '.join('{} {:.3f}'.format(k, v * 50000 / sub) for k, v in ms.items()), 'sum per 50k {:.3f}'.format(sum(ms.values()) * 50000 / sub))
