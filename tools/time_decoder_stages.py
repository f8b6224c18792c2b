"""Per-kernel HIP-event times of the decoder on one 50000-query band chunk, fp32 and f16x3.  usage: python tools/time_decoder_stages.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench_workloads as workloads
from ppsurf_amd.decoder import DecoderPlan, ChunkPipeline
from ppsurf_amd.synthetic import make_cloud, make_latents, network_state_dict

DEV = 'cuda:0'
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sd = network_state_dict('ppsurf')
cloud = make_cloud(100000, seed=42)
pts = torch.from_numpy(cloud).to(DEV)
lat = torch.from_numpy(make_latents(256, 100000, seed=77)[0]).to(DEV)
chunks, _ = workloads.band_chunks(cloud, 257, 50000, DEV)
names = ('interp_pool', 'pointnet_stn_rows', 'pointnet_stn_fc', 'pointnet_feat_rows', 'decode_tail')
for dtype in ('f32', 'f16x3'):
    plan = DecoderPlan(sd, DEV, dtype=dtype)
    pipe = ChunkPipeline(plan, plan.point_table(lat), pts, pts, 64, 50, same_cloud=True, max_chunk=50000)
    ev = [workloads.HipEvents(6) for _ in range(reps)]
    for c in chunks[:3]:
        pipe.run([c])
    torch.cuda.synchronize()
    for i in range(reps):
        pipe.run([chunks[3 + i % (len(chunks) - 3)]], stage_events=[ev[i].arr])
    torch.cuda.synchronize()
    ms = {n: float(np.median([e.elapsed_ms(j, j + 1) for e in ev])) for j, n in enumerate(names)}
    print(dtype, ' '.join('{} {:.3f}'.format(k, v) for k, v in ms.items()), 'sum {:.3f} ms'.format(sum(ms.values())))
