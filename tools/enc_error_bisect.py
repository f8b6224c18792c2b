"""Where does the encoder's deviation from the reference enter?  (end-to-end 2e-4 instead of 1e-4.)

For every stage of FKAConvNetwork the HIP kernels are fed the ORACLE's fp32 input of that stage and compared with the
oracle evaluated in float64 on the same input (local error), next to the oracle's own fp32-vs-fp64 deviation (the
reference's noise floor) and the accumulated end-to-end error.   usage: python tools/enc_error_bisect.py [N]
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import ppsurf_oracle as O                                   # noqa: E402  (measurement tool, not product)
from ppsurf_amd.encoder import EncoderPlan                               # noqa: E402
from ppsurf_amd.synthetic import network_state_dict, make_cloud          # noqa: E402

DEV = 'cuda:0'
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
torch.set_num_threads(16)
sd = network_state_dict('ppsurf')
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
rng = np.random.default_rng(3)
cloud = make_cloud(N, seed=4)
pts = torch.from_numpy(cloud.T.copy()).unsqueeze(0)
sups, cur = [], pts
for _ in range(4):
    sel = torch.from_numpy(np.sort(rng.choice(cur.shape[2], max(1, int(cur.shape[2] * 0.25)), replace=False)))
    cur = cur[:, :, sel].contiguous()
    sups.append(cur)
data = {'pts': pts}
data.update(O.fkaconv_ids_from_supports(pts, sups))

calls = []
_rb, _fka = O.residual_block, O.fkaconv_layer


def rb(sd_, p, x, pts_, sup_, ids_, act='relu'):
    out = _rb(sd_, p, x, pts_, sup_, ids_, act)
    if out.dtype == torch.float32:
        calls.append(('rb', p, x, pts_, sup_, ids_, out))
    return out


def fka(sd_, p, x, pts_, sup_, ids_, act='relu'):
    out = _fka(sd_, p, x, pts_, sup_, ids_, act)
    if out.dtype == torch.float32:
        calls.append(('fka', p, x, pts_, sup_, ids_, out))
    return out


O.residual_block, O.fkaconv_layer = rb, fka
with torch.no_grad():
    lat32 = O.fkaconv_network(sd, 'encoder', data, act='silu', fixed=True)
O.residual_block, O.fkaconv_layer = _rb, _fka

plan = EncoderPlan({k: v for k, v in sd.items()}, DEV, prefix='encoder', act='silu', fixed=True)
pm = lambda t: t[0].t().contiguous().to(DEV)
d = lambda t: t.double()


def report(name, hip, ref64, ref32):
    hip = hip.double().cpu()
    s = ref64.abs().max().item()
    print('{:34s} scale {:8.2e} | HIP-f64 {:9.2e} ({:8.1e} rel) | ref32-f64 {:9.2e} | rows {}'.format(
        name, s, (hip - ref64).abs().max().item(), (hip - ref64).abs().max().item() / s, (ref32.double() - ref64).abs().max().item(),
        tuple(ref64.shape)))


with torch.no_grad():
    for kind, p, x, pts_, sup_, ids_, out32 in calls:
        xg, pg, sg, ig = pm(x), pm(pts_), pm(sup_), ids_[0].contiguous().to(DEV)
        if kind == 'fka' and p == 'encoder.cv0':
            ref64 = torch.relu(O._bn(sd64, 'encoder.bn0', _fka(sd64, p, d(x), d(pts_), d(sup_), ids_, 'silu')))
            ref32 = torch.relu(O._bn(sd, 'encoder.bn0', out32))
            report('cv0+bn0+relu', plan.cv0(xg, pg, sg, ig).t(), ref64[0], ref32[0])
        elif kind == 'rb':
            name = p.split('.')[-1]
            blk = plan.blocks[name[len('resnetb'):]]
            # stage 1: cv0 + bn0 + relu
            h64 = torch.relu(O._bn(sd64, p + '.bn0', O._conv1(sd64, p + '.cv0', d(x))))
            h32 = torch.relu(O._bn(sd, p + '.bn0', O._conv1(sd, p + '.cv0', x)))
            hh = blk.cv0(xg, relu=True)
            report(name + '.cv0(lin)', hh.t(), h64[0], h32[0])
            # stage 2: FKAConv + bn1 + relu, fed the oracle's fp32 h
            f64 = torch.relu(O._bn(sd64, p + '.bn1', _fka(sd64, p + '.cv1', d(h32), d(pts_), d(sup_), ids_, 'silu')))
            f32 = torch.relu(O._bn(sd, p + '.bn1', _fka(sd, p + '.cv1', h32, pts_, sup_, ids_, 'silu')))
            hf = blk.cv1(pm(h32), pg, sg, ig)
            report(name + '.cv1(fka K={})'.format(16 * h32.shape[1]), hf.t(), f64[0], f32[0])
            # whole block
            report(name + ' (block)', blk(xg, pg, sg, ig).t(), _rb(sd64, p, d(x), d(pts_), d(sup_), ids_, 'silu')[0], out32[0])
    # end to end
    ids = {}
    for k, v in data.items():
        if k.startswith('ids'):
            t = v[0].to(DEV)
            ids[k] = t.reshape(-1).contiguous() if k in ('ids43', 'ids32', 'ids21', 'ids10') else t.contiguous()
    lat = plan.forward(pm(pts), [pm(s) for s in sups], ids)
    d64 = {k: (d(v) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in data.items()}
    lat64 = O.fkaconv_network(sd64, 'encoder', d64, act='silu', fixed=True)
    report('END-TO-END latents', lat.t(), lat64[0], lat32[0])
    print('HIP vs ref32 end-to-end max abs', (lat.t().cpu() - lat32[0]).abs().max().item())
