"""Per-kernel HIP-event times of the fp32 decoder for chunk sizes around rec_batch_size: does the time per query depend on how the chunk divides
into the persistent kernels' tiles?  usage: python tools/time_decoder_q.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_workloads as workloads          # noqa: E402
from ppsurf_amd.decoder import DecoderPlan, ChunkPipeline          # noqa: E402
from ppsurf_amd.synthetic import make_cloud, make_latents, network_state_dict          # noqa: E402

DEV = 'cuda:0'
sd = network_state_dict('ppsurf')
cloud = make_cloud(100000, seed=42)
pts = torch.from_numpy(cloud).to(DEV)
lat = torch.from_numpy(make_latents(256, 100000, seed=77)[0]).to(DEV)
chunks, _ = workloads.band_chunks(cloud, 257, 50000, DEV)
big = torch.cat(chunks[:3])
names = ('interp_pool', 'pointnet_stn_rows', 'pointnet_stn_fc', 'pointnet_feat_rows', 'decode_tail')
plan = DecoderPlan(sd, DEV, dtype='f32')
pipe = ChunkPipeline(plan, plan.point_table(lat), pts, pts, 64, 50, same_cloud=True, max_chunk=70000)
for q in (32768, 49152, 50000, 51200, 65536):
    c = big[:q].contiguous()
    ev = [workloads.HipEvents(6) for _ in range(12)]
    for _ in range(2):
        pipe.run([c])
    torch.cuda.synchronize()
    for e in ev:
        pipe.run([c], stage_events=[e.arr])
    torch.cuda.synchronize()
    ms = {n: float(np.median([e.elapsed_ms(j, j + 1) for e in ev])) for j, n in enumerate(names)}
    print('Q {:6d}: '.format(q) + ' '.join('{} {:.3f}'.format(k, v) for k, v in ms.items()) + '  | us per 1000 queries: '
          + ' '.join('{:.1f}'.format(v / q * 1e6) for v in ms.values()))
