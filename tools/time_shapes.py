"""Several whole R = 257 reconstructions in a row by the product driver (bench_workloads.reconstruct_steered): seconds of the latent loop and of the
surface extraction per shape -- run-to-run spread of the shapes/hour leg of bench.py.   python tools/time_shapes.py [shapes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_workloads as workloads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
model = workloads.make_model(device='cuda:0')
for i in range(n):
    r = workloads.reconstruct_steered(model, 100000, seed=42 + i, device='cuda:0')
    print('shape {}: latent loop {:.3f} s, surface {:.3f} s, total {:.3f} s ({:.0f} shapes/hour), {} decoder queries, reserved {:.1f} GB'.format(
        i, r['latent_s'], r['surface_s'], r['total_s'], 3600 / r['total_s'], r['decoder_queries'], torch.cuda.memory_reserved() / 1e9), flush=True)
