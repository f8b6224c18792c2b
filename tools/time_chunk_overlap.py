"""Does running the spatial stage (kNN + patches: vector ALU) of chunk i+1 on a side stream underneath the decoder kernels (matrix pipe) of chunk i pay?
ChunkPipeline(overlap=False | True) over the band chunks of one shape, whole lists per run() call.   python tools/time_chunk_overlap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import bench_workloads as workloads
from ppsurf_amd.decoder import ChunkPipeline
from ppsurf_amd.synthetic import make_cloud, make_latents

DEV = 'cuda:0'
model = workloads.make_model(device=DEV)
plan = model.network.decoder_plan(DEV)
cloud = make_cloud(100000, seed=42)
pts = torch.from_numpy(cloud).to(DEV)
table = plan.point_table(torch.from_numpy(make_latents(256, 100000, seed=77)[0]).to(DEV))
chunks, _ = workloads.band_chunks(cloud, 257, 50000, DEV)
chunks = chunks[:40]
for rep in range(2):
    for ov in (False, True):
        pipe = ChunkPipeline(plan, table, pts, pts, 64, 50, same_cloud=True, max_chunk=50000, overlap=ov)
        pipe.run(chunks[:4])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = pipe.run(chunks)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('overlap={}: {:.3f} ms per chunk, {:.2f} M q/s'.format(ov, dt / len(chunks) * 1e3, sum(c.shape[0] for c in chunks) / dt / 1e6))

# two independent chunk streams of one process on two HIP streams (what two ranks sharing a GPU do: +3 % in profiles/round4_bench_2rank_rehearsal.json)
from ppsurf_amd.decoder import DecoderPlan
sd = {k: v for k, v in model.network.state_dict().items()}
plans = [DecoderPlan(sd, DEV), DecoderPlan(sd, DEV)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
pipes = []
for pl, st in zip(plans, streams):
    with torch.cuda.stream(st):
        pipes.append(ChunkPipeline(pl, pl.point_table(torch.from_numpy(make_latents(256, 100000, seed=77)[0]).to(DEV)), pts, pts, 64, 50, same_cloud=True, max_chunk=50000))
torch.cuda.synchronize()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i, c in enumerate(chunks):
        with torch.cuda.stream(streams[i & 1]):
            pipes[i & 1].run([c])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('two streams: {:.3f} ms per chunk, {:.2f} M q/s'.format(dt / len(chunks) * 1e3, sum(c.shape[0] for c in chunks) / dt / 1e6))
