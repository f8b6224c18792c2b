"""Fit step time by module, by ablation: the step of workloads.FitStep timed whole, then with the PointNet branch / the interpolation
head replaced by a zero that keeps the autograd graph connected (the encoder still gets a backward pass), then data preparation alone.
Usage: python tools/fit_module_breakdown.py [--steps 10]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppsurf_amd import spatial, train_graph          # noqa: E402
import bench_workloads as workloads


def timed(step, n):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    a = ap.parse_args()
    step = workloads.FitStep(overlap_prep=False)          # one thread, one stream: attributable timings
    full = timed(step, a.steps)
    real_pn, real_ia = train_graph.pointnet, train_graph.interp_attention

    def no_pn(pn, patches, need_trans=True):
        return torch.zeros((patches.shape[0], 256), device=patches.device, dtype=torch.float32), None

    def no_ia(proj, latents, pts, query, ids, last_layer=True):
        b, q = query.shape[0], query.shape[1]
        return (latents.sum() * 0).expand(b, q, 256)

    train_graph.pointnet = no_pn
    t_nopn = timed(step, a.steps)
    train_graph.interp_attention = no_ia
    t_enc = timed(step, a.steps)
    train_graph.pointnet = real_pn
    t_noia = timed(step, a.steps)
    train_graph.interp_attention = real_ia

    def prep():
        batch = dict(step.batches[0])
        b = batch['pts_ms'].shape[0]
        batch['pts_local_ps'] = spatial.get_pts_local_ps_batch([batch['pts_ms'][i] for i in range(b)], batch['pts_query_ms'], step.p)
        spatial.get_data_poco(batch)
    t_prep = timed(prep, a.steps)
    print('full {:.2f} ms | without PointNet {:.2f} (PointNet {:.2f}) | without interpolation head {:.2f} (head {:.2f}) | '
          'encoder + MLP + prep + AdamW {:.2f} | data preparation alone {:.2f}'.format(full, t_nopn, full - t_nopn, t_noia, full - t_noia, t_enc, t_prep))


if __name__ == '__main__':
    main()
