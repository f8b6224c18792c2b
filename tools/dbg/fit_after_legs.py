"""Which leg of the default bench line makes the in-process fit leg slow (21.5 ms instead of 19.7)?   python tools/dbg/fit_after_legs.py --legs head,f32,dense,shapes,config2,config5
Runs the named legs with the bench's own functions at the default sizes, then times a new FitStep like tools/dbg/fit_after_history.py."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench                                   # noqa: E402
import bench_workloads as workloads            # noqa: E402
from fit_after_history import timed_fit        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--legs', default='head,f32,dense,shapes,config2,config5')
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--dummies', type=int, default=0, help='torch.cuda.Stream() objects created (and used once) right before the FitStep')
    a = ap.parse_args()
    legs = set(a.legs.split(',')) if a.legs else set()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    from ppsurf_amd.decoder import DecoderPlan, ChunkPipeline
    from ppsurf_amd.synthetic import network_state_dict
    sd = network_state_dict('ppsurf')
    plan = DecoderPlan(sd, dev, dtype='f16x3')
    if legs & {'head', 'f32', 'dense'}:
        shapes, work = bench.build_work(plan, 10 + a.steps, 0, dev)
        if 'head' in legs:
            bench.chunk_loop(work, a.steps, 10, 0, 1, None, dev)
        if 'f32' in legs:
            plan2 = DecoderPlan(sd, dev, dtype='f32')
            pipes2, work2 = {}, []
            for pipe, c in work:
                if id(pipe) not in pipes2:
                    pipes2[id(pipe)] = ChunkPipeline(plan2, pipe.table, pipe.pts, pipe.pts, 64, 50, same_cloud=True, max_chunk=50000)
                work2.append((pipes2[id(pipe)], c))
            bench.chunk_loop(work2, min(a.steps, 100), 5, 0, 1, None, dev, min_timed_s=1.0)
            del plan2, pipes2, work2
        if 'dense' in legs:
            _, workd = bench.build_work(plan, 45, 0, dev, queries='dense')
            bench.chunk_loop(workd, 40, 5, 0, 1, None, dev, min_timed_s=0.5)
            del workd
        del work, shapes
        torch.cuda.empty_cache()
    if 'shapes' in legs:
        model = workloads.make_model(257, 50, 50000, dev)
        for i in range(3):
            workloads.reconstruct_steered(model, 100000, seed=42 + i, device=dev)
        model.network.decoder_dtype = 'f32'
        for i in range(3):
            workloads.reconstruct_steered(model, 100000, seed=42 + i, device=dev)
        del model
        torch.cuda.empty_cache()
    if 'config2' in legs:
        bench.config2_leg(dev, 'f16x3')
        torch.cuda.empty_cache()
    if 'config5' in legs:
        bench.config5_leg(dev, 'f16x3')
        torch.cuda.empty_cache()
    del plan
    keep = []
    x = torch.zeros(1024, device=dev)
    for _ in range(a.dummies):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            x.add_(1.0)
        keep.append(st)
    torch.cuda.synchronize()
    timed_fit('dummies {} '.format(a.dummies) + 'after legs [{}] steps {}'.format(a.legs, a.steps))


if __name__ == '__main__':
    main()
