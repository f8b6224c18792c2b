"""The interpolation head's forward chain at the fit batch's size (20 000 queries x 64 neighbours, 100 000-row table): the one-kernel chain
(pps_head_chain_fwd + pps_attn_pool_fwd) against the separate launches, HIP events.   python tools/time_head_chain.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from ppsurf_amd import train_ops  # noqa: E402
import test_gpu_head_chain as T   # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dt = torch.bfloat16
    table, ids, pts, query, wx, w2, b2, w3, b3, wq, bq = T._case(20000, 64, 100000, 1, dt)
    # neighbour ids of a query are close to each other in a real batch; random ids are the worst case for the gather
    with torch.no_grad(), torch.autocast('cuda', dtype=dt):
        print('chain kernel + attention pooling: {:.3f} ms'.format(timed(lambda: train_ops.head_chain(table, ids, pts, query, 64, wx, (w2, b2), (w3, b3), (wq, bq)))))
        print('separate launches              : {:.3f} ms'.format(timed(lambda: T._separate(table, ids, pts, query, 64, wx, w2, b2, w3, b3, wq, bq))))
        y3 = torch.empty((20000 * 64, 256), device='cuda', dtype=dt).normal_()
        qy = torch.empty((20000 * 64, 64), device='cuda', dtype=dt).normal_()
        print('attention pooling alone        : {:.3f} ms'.format(timed(lambda: train_ops.attn_pool(qy.view(20000, 64, 64), y3.view(20000, 64, 256), True))))


if __name__ == '__main__':
    main()
