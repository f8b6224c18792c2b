"""Where a fit step spends its GPU time, by network part (forward+backward of each part timed in isolation, bf16-mixed).
    python tools/train_breakdown.py [--fp32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from ppsurf_amd import modules, train_graph as tg  # noqa: E402
from golden_util import filled_sd  # noqa: E402
from time_train_step import make_batch, prepare  # noqa: E402


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--fp32', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda')
    net = modules.PPSurfNetwork(in_channels=3, latent_size=256, out_channels=2, k=64, num_pts_local=50, pointnet_latent_size=256)
    net.load_state_dict(filled_sd('', 'ppsurf'))
    net = net.to(dev).train()
    batch = prepare(make_batch(10, 10000, 2000, 50, dev), 50)
    ac = lambda: torch.autocast('cuda', dtype=torch.bfloat16, enabled=not a.fp32)
    pm = lambda t: t.transpose(1, 2).contiguous()
    lat = torch.randn(10, 10000, 256, device=dev, requires_grad=True)
    pts, q = pm(batch['pts']), pm(batch['pts_query'])
    patches = batch['pts_local_ps'].reshape(-1, 50, 3)
    feat = torch.randn(20000, 256, device=dev, requires_grad=True)

    def run(f):
        def g():
            net.zero_grad(set_to_none=True)
            with ac():
                out = f()
            out.float().square().mean().backward()
        return g

    def fwd_only(f):
        def g():
            with torch.no_grad(), ac():
                f()
        return g

    parts = {'encoder': lambda: tg.encoder(net.encoder, batch),
             'interp attention': lambda: tg.interp_attention(net.projection, lat, pts, q, batch['proj_ids']),
             'pointnet': lambda: tg.pointnet(net.point_net, patches)[0],
             'mlp': lambda: tg.mlp(net.mlp, feat)}
    for name, f in parts.items():
        print('{:18s} fwd {:7.2f} ms   fwd+bwd {:7.2f} ms'.format(name, timed(fwd_only(f)), timed(run(f))))
    x0 = torch.randn(10, 10000, 64, device=dev, requires_grad=True)
    blk = net.encoder.resnetb01
    print('{:18s} fwd {:7.2f} ms   fwd+bwd {:7.2f} ms'.format('  resnetb01 alone', timed(fwd_only(lambda: tg.residual_block(blk, x0, pts, pts, batch['ids00']))),
                                                            timed(run(lambda: tg.residual_block(blk, x0, pts, pts, batch['ids00'])))))
    h = torch.randn(10, 10000, 32, device=dev, requires_grad=True)
    print('{:18s} fwd {:7.2f} ms   fwd+bwd {:7.2f} ms'.format('  its FKAConv', timed(fwd_only(lambda: tg.fkaconv_layer(blk.cv1, h, pts, pts, batch['ids00']))),
                                                            timed(run(lambda: tg.fkaconv_layer(blk.cv1, h, pts, pts, batch['ids00'])))))


if __name__ == '__main__':
    main()
