"""Per-shape times of the training step's dense-layer kernels (csrc/pps_gemm_train.hip) against the library GEMM of the same product, for every
layer shape of a fit batch (B = 10 x 10 000 points, 2000 queries; source/base/nn.py:438-554, source/poco_model.py:405-417).
    python tools/time_gemm_shapes.py            -> table on stdout (us per call: forward NT, input-gradient NT, weight-gradient TN; own | library)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppsurf_amd import train_ops  # noqa: E402

R = [100000, 25000, 6250, 1560, 390]
H = 64
SHAPES = []           # (name, rows, K, N)


def block(name, cin, cout, r_in, r_out):
    half = cin // 2
    SHAPES.append((name + '.cv0', r_in, cin, half))
    SHAPES.append((name + '.cv1', r_out, half * 16, half))
    SHAPES.append((name + '.cv2', r_out, half, cout))
    if cin != cout:
        SHAPES.append((name + '.sc', r_in, cin, cout))


SHAPES.append(('cv0', R[0], 48, H))
block('b01', H, H, R[0], R[0])
block('b10', H, 2 * H, R[0], R[1])
block('b11', 2 * H, 2 * H, R[1], R[1])
block('b20', 2 * H, 4 * H, R[1], R[2])
block('b21', 4 * H, 4 * H, R[2], R[2])
block('b30', 4 * H, 8 * H, R[2], R[3])
block('b31', 8 * H, 8 * H, R[3], R[3])
block('b40', 8 * H, 16 * H, R[3], R[4])
block('b41', 16 * H, 16 * H, R[4], R[4])
SHAPES += [('cv5', R[4], 32 * H, 16 * H), ('cv3d', R[3], 24 * H, 8 * H), ('cv2d', R[2], 12 * H, 4 * H), ('cv1d', R[1], 6 * H, 2 * H),
           ('cv0d', R[0], 3 * H, H), ('fcout', R[0], H, 256), ('fc1.table', R[0], 256, 256), ('fc_value', 20000, 256, 256), ('fc8', 20000, 256, 256),
           ('stn.fc1', 20000, 256, 128), ('stn.fc2', 20000, 128, 64), ('stn.fc3', 20000, 64, 4096), ('att.fc_value', 20000, 256, 256),
           ('mlp.0', 20000, 256, 256), ('mlp.1', 20000, 256, 256), ('mlp.2', 20000, 256, 8)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1000.0


def main():
    dev = 'cuda:0'
    dt = torch.bfloat16
    tot = [0.0] * 6
    print('{:14s} {:>7s} {:>5s} {:>5s} | {:>8s} {:>8s} | {:>8s} {:>8s} | {:>8s} {:>8s}'.format('layer', 'rows', 'K', 'N', 'fwd', 'lib', 'dx', 'lib', 'dw', 'lib'))
    for name, rows, k, n in SHAPES:
        x = torch.randn(rows, k, device=dev).to(dt)
        w = torch.randn(n, k, device=dev).to(dt)
        wt = w.t().contiguous()
        g = torch.randn(rows, n, device=dev).to(dt)
        bias = torch.randn(n, device=dev)
        t = [timed(lambda: train_ops.gemm_nt(x, w, bias)), timed(lambda: torch.nn.functional.linear(x, w, bias.to(dt))),
             timed(lambda: train_ops.gemm_nt(g, wt)), timed(lambda: g @ w),
             timed(lambda: train_ops.gemm_tn(g, x)), timed(lambda: torch.mm(g.t(), x, out_dtype=torch.float32))]
        for i in range(6):
            tot[i] += t[i]
        extra = ''
        if train_ops.rows_layer_supported(rows, k, n):          # the LDS-resident-weight row kernels: forward | backward (dx + dw + db in one call)
            w32 = w.float().requires_grad_(True)
            b32 = bias.clone().requires_grad_(True)
            xg = x.clone().requires_grad_(True)
            with torch.autocast('cuda', dtype=dt):
                y = train_ops.rows_layer(train_ops.Act(xg), w32, b32).raw
            t_f = timed(lambda: train_ops.rows_layer(train_ops.Act(x), w32.detach(), b32.detach()))
            t_b = timed(lambda: torch.autograd.grad(y, (xg, w32, b32), g, retain_graph=True))
            extra = ' | rows_layer fwd {:6.1f} bwd(dx+dw+db) {:6.1f}'.format(t_f, t_b)
        print('{:14s} {:7d} {:5d} {:5d} | {:8.1f} {:8.1f} | {:8.1f} {:8.1f} | {:8.1f} {:8.1f}{}'.format(name, rows, k, n, *t, extra), flush=True)
    print('{:34s} | {:8.1f} {:8.1f} | {:8.1f} {:8.1f} | {:8.1f} {:8.1f}'.format('sum (us)', *tot))


if __name__ == '__main__':
    main()
