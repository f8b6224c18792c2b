"""Fused dense layer of the training step (train_ops.rows_layer) against the separate ops it replaces (library GEMM + split-K weight gradient
+ the fused BatchNorm/ReLU op), forward and forward + backward, at the row counts of BASELINE config 3 (1 M patch points, 1.28 M neighbours).
Usage: python tools/time_rows_layer.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppsurf_amd import train_graph, train_ops          # noqa: E402

DEV = 'cuda:0'


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


class BN:
    def __init__(self, c):
        self.weight = torch.ones(c, device=DEV, requires_grad=True)
        self.bias = torch.zeros(c, device=DEV, requires_grad=True)
        self.running_mean, self.running_var = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        self.momentum, self.eps, self.training, self.track_running_stats, self.num_batches_tracked = 0.1, 1e-5, True, True, None


def main():
    print('rows cin cout bn | fused fwd / fwd+bwd ms (GB/s of the minimal traffic) | separate ops fwd / fwd+bwd ms')
    for rows, cin, cout, bn in [(1000000, 64, 64, True), (1000000, 64, 128, True), (1000000, 128, 256, True), (1280000, 256, 256, False),
                                (1280000, 256, 64, False)]:
        x = torch.randn(rows, cin, device=DEV).to(torch.bfloat16)
        w = (torch.randn(cout, cin, device=DEV) / cin ** 0.5).requires_grad_(True)
        b = torch.zeros(cout, device=DEV, requires_grad=True)
        aff = torch.stack([torch.ones(cin, device=DEV), torch.zeros(cin, device=DEV)]).requires_grad_(True)
        gy = torch.randn(rows, cout, device=DEV).to(torch.bfloat16)
        hold = BN(cout) if bn else None

        def fused(backward):
            xi = x.detach().requires_grad_(True)
            out = train_ops.rows_layer(train_ops.Act(xi, aff, True), w, b, hold, relu=True)
            if backward:
                torch.autograd.backward([out.raw] + ([out.affine] if bn else []), [gy] + ([torch.ones_like(out.affine)] if bn else []))

        def separate(backward):
            xi = x.detach().requires_grad_(True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                a = torch.relu(xi)                                   # stands for the materialised activation of the previous layer (its cost is in the old bn_act)
                y = train_graph.rows_linear(a, w, b)
                if bn:
                    y = train_ops.bn_act(y, hold.weight, hold.bias, hold.running_mean, hold.running_var, 0.1, 1e-5, True)
            if backward:
                y.backward(gy)

        f0, f1 = timed(lambda: fused(False)), timed(lambda: fused(True))
        s0, s1 = timed(lambda: separate(False)), timed(lambda: separate(True))
        fb = rows * (cin + cout) * 2
        bb = rows * (2 * ((2 if bn else 1) * cout + cin) + cin) * 2
        print('{} {} {} {} | {:.3f} ({:.0f}) / {:.3f} (bwd {:.0f}) | {:.3f} / {:.3f}'.format(rows, cin, cout, int(bn), f0, fb / f0 / 1e6, f1,
                                                                                 bb / max(f1 - f0, 1e-9) / 1e6, s0, s1))


if __name__ == '__main__':
    main()
