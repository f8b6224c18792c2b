"""BASELINE config 5 (ppsurf_200nn, R=513, 250k-point synthetic cloud, rec_batch_size 25000): queries/s of the chunk loop.
    python tools/time_config5.py [--p 200] [--n 250000] [--q 25000] [--steps 10]"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--p', type=int, default=200)
    ap.add_argument('--n', type=int, default=250000)
    ap.add_argument('--q', type=int, default=25000)
    ap.add_argument('--res', type=int, default=513)
    ap.add_argument('--steps', type=int, default=10)
    a = ap.parse_args()
    from golden_util import filled_sd
    from ppsurf_amd.decoder import DecoderPlan, ChunkPipeline
    from ppsurf_amd.synthetic import make_cloud, make_band_queries, make_latents
    dev = torch.device('cuda', 0)
    plan = DecoderPlan(filled_sd('', key='ppsurf'), dev)
    cloud = make_cloud(a.n, seed=42)
    qd = torch.from_numpy(make_band_queries(cloud, a.q, resolution=a.res, seed=1)).to(dev)
    pts = torch.from_numpy(cloud).to(dev)
    table = plan.point_table(torch.from_numpy(make_latents(256, a.n, seed=77)[0]).to(dev))
    pipe = ChunkPipeline(plan, table, pts, pts, 64, a.p, same_cloud=True, max_chunk=a.q)
    pipe.run([qd] * 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = pipe.run([qd] * a.steps, want_occ=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert bool(torch.isfinite(res[-1][1]).all())
    from bench_workloads import HipEvents
    evs = [HipEvents(6) for _ in range(a.steps)]
    pipe.run([qd] * a.steps, want_occ=True, stage_events=[e.arr for e in evs])
    torch.cuda.synchronize()
    names = ['interp_pool', 'pointnet_stn_rows', 'pointnet_stn_fc', 'pointnet_feat_rows', 'decode_tail']
    stage = [sum(e.elapsed_ms(i, i + 1) for e in evs) / a.steps for i in range(5)]
    print('decoder stages (HIP events, ms): ' + ', '.join('{} {:.3f}'.format(n, t) for n, t in zip(names, stage)) +
          ' | sum {:.2f} of {:.2f} ms/step (the rest: kNN + patches)'.format(sum(stage), dt / a.steps * 1e3))
    mflop = {50: 53.21, 200: 102.50}.get(a.p)
    print('P={} N={} Q={} R={}: {:.2f} ms/step, {:.3f} M queries/s{}'.format(
        a.p, a.n, a.q, a.res, dt / a.steps * 1e3, a.q * a.steps / dt / 1e6,
        ' ({:.1f} algorithmic TFLOP/s at {} MFLOP/query)'.format(mflop * a.q * a.steps / dt / 1e6, mflop) if mflop else ''))


if __name__ == '__main__':
    main()
