"""Which framework ops the fit step's device time belongs to: torch.profiler over a few steps of workloads.FitStep, device time
summed per aten / autograd op (self time), so that the long tail of element-wise / reduction kernels of the rocprofv3 summary can
be attributed to the layers that launch them.  Usage: python tools/fit_op_profile.py [--steps 3] [--rows 45]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_workloads as workloads          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--rows', type=int, default=45)
    ap.add_argument('--encoder-only', action='store_true', help='PointNet and interpolation head replaced by zeros (tools/fit_module_breakdown.py): encoder + MLP + data preparation + AdamW')
    ap.add_argument('--kernels', action='store_true', help='list device kernels instead of framework ops')
    a = ap.parse_args()
    step = workloads.FitStep(overlap_prep=False)          # one thread, one stream: attributable timings
    if a.encoder_only:
        from ppsurf_amd import train_graph
        train_graph.pointnet = lambda pn, patches, need_trans=True: (torch.zeros((patches.shape[0], 256), device=patches.device), None)
        train_graph.interp_attention = lambda proj, latents, pts, query, ids, last_layer=True: (latents.sum() * 0).expand(query.shape[0], query.shape[1], 256)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=not a.kernels)
    rows = []
    for e in ka:
        t = getattr(e, 'self_device_time_total', None)
        if t is None:
            t = e.self_cuda_time_total
        if t > 0 and (not a.kernels or e.device_type.name != 'CPU'):
            rows.append((t / a.steps / 1000.0, e.count / a.steps, e.key, str(e.input_shapes)[:90]))
    rows.sort(reverse=True)
    print('total device ms/step {:.2f}'.format(sum(r[0] for r in rows)))
    for t, c, k, s in rows[:a.rows]:
        print('{:8.3f} ms {:6.1f} calls  {:<44s} {}'.format(t, c, k[:44], s))


if __name__ == '__main__':
    main()
