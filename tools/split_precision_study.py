"""CPU study behind the opt-in decoder dtype "f16x3" (csrc/pps_common.h): replays the decoder pipeline of tests/emulate.py in float64
with every dense product replaced by an emulated split-precision product -- operands rounded to (hi, lo) pairs of f16 (round to
nearest, or toward zero like v_cvt_pkrtz_f16_f32) or bf16, 1 / 3 / 4 partial products, exact accumulation -- and reports the max
logit error on the reference's golden case at two latent magnitudes.  Result (this container): f16 x 3 products 8e-7 / 1.1e-5
(round toward zero 1.0e-6 / 1.2e-5), bf16 x 3 7e-6 / 2.8e-4 (fails the 1e-4 bar at the magnitude real latents have), single f16
product 5e-4 / 1.3e-2.        python tools/split_precision_study.py
"""
import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import emulate
from golden_util import load_golden, filled_sd
from ppsurf_amd.decoder import DecoderPlan
from ppsurf_amd.synthetic import make_latents

def rnd(x, kind):
    t = torch.from_numpy(np.asarray(x, dtype=np.float32))
    if kind=='f16': return t.to(torch.float16).to(torch.float64).numpy()
    if kind=='bf16': return t.to(torch.bfloat16).to(torch.float64).numpy()
def rtz16(x32):
    h = x32.astype(np.float16)
    over = np.abs(h.astype(np.float32)) > np.abs(x32)
    h2 = np.nextafter(h, np.float16(0))
    return np.where(over, h2, h).astype(np.float64)
def split(x, kind, rtz=False):
    x32 = np.asarray(x, dtype=np.float32)
    if rtz:
        hi = rtz16(x32)
        lo = rtz16((x32.astype(np.float64)-hi).astype(np.float32))
        return hi, lo
    hi = rnd(x32, kind)
    lo = rnd((x32.astype(np.float64)-hi).astype(np.float32), kind)
    return hi, lo
MODE=None
stats={}
def mm(x, wT):   # x [...,K] @ wT [K,N]
    if MODE is None: return x @ wT
    kind, nprod = MODE
    xh, xl = split(x, kind, rtz=(kind=='f16rtz')); wh, wl = split(wT, 'f16' if kind=='f16rtz' else kind)
    stats['xmax']=max(stats.get('xmax',0), float(np.abs(x).max()))
    y = xh @ wh
    if nprod>=2: y = y + xh @ wl
    if nprod>=3: y = y + xl @ wh
    if nprod>=4: y = y + xl @ wl
    return y.astype(np.float32).astype(np.float64)

# monkeypatch emulate.decode's matmuls: re-implement decode with mm for the dense layers
def decode(w, latents_cn, pts, query, idx, patches):
    E=emulate
    g_w = E.unpack_dense(w['g_w'], 256, 256)
    G = (latents_cn.T.astype(np.float64) @ g_w.T + w['g_b']).astype(np.float32).astype(np.float64)   # per-shape table stays fp32
    xyz, fc2, fc3, fcq = E._split(w['ip_w'], [1024, 65536, 65536, 16384])
    b2, b3, bq = E._split(w['ip_b'].astype(np.float64), [256, 256, 64])
    rel = query[:, None, :].astype(np.float64) - pts[idx]
    h = E.relu(G[idx] + rel @ E.unpack_xyz(xyz, 256).T)          # xyz part on VALU fp32
    h = E.relu(mm(h, E.unpack_dense(fc2, 256, 256).T) + b2)
    h = E.relu(mm(h, E.unpack_dense(fc3, 256, 256).T) + b3)
    att = E.softmax(mm(h, E.unpack_dense(fcq, 64, 256).T) + bq, axis=1).mean(axis=2)
    pooled = (att[:, :, None] * h).sum(axis=1)
    xa, c0b, s1, s2, s3 = E._split(w['pa_w'], [256, 4096, 4096, 8192, 32768])
    ba = E._split(w['pa_b'].astype(np.float64), [64, 64, 64, 128, 256])
    x = patches.astype(np.float64)
    x0 = E.relu(x @ E.unpack_xyz(xa, 64).T + ba[0])
    x1 = E.relu(mm(x0, E.unpack_dense(c0b, 64, 64).T) + ba[1])
    t = E.relu(mm(x1, E.unpack_dense(s1, 64, 64).T) + ba[2])
    t = E.relu(mm(t, E.unpack_dense(s2, 128, 64).T) + ba[3])
    t = E.relu(mm(t, E.unpack_dense(s3, 256, 128).T) + ba[4])
    gmax = t.max(axis=1)
    f1, f2, f3 = E._split(w['pb_w'], [32768, 8192, 262144])
    bb = E._split(w['pb_b'].astype(np.float64), [128, 64, 4096])
    u = E.relu(gmax @ E.unpack_dense(f1, 128, 256).T + bb[0])          # per-query small layers stay fp32
    u = E.relu(u @ E.unpack_dense(f2, 64, 128).T + bb[1])
    trans2 = (u @ E.unpack_dense(f3, 4096, 64).T + bb[2]).reshape(-1, 64, 64)
    xc, c0b2, c2 = E._split(w['pc_w'], [256, 4096, 8192])      # round 4: conv1 lives in the per-query matrix, conv3 in the tail (decoder.py)
    bc = E._split(w['pc_b'].astype(np.float64), [64, 64, 64, 128, 256, 256, 4])
    y0 = E.relu(x @ E.unpack_xyz(xc, 64).T + bc[0])
    y1 = E.relu(mm(y0, E.unpack_dense(c0b2, 64, 64).T) + bc[1])
    if MODE is None:
        y = np.einsum('qab,qpb->qpa', trans2, y1)
    else:
        y = np.stack([mm(y1[q], trans2[q].T) for q in range(y1.shape[0])])
    y = E.relu(y + bc[2])
    y = E.relu(mm(y, E.unpack_dense(c2, 128, 64).T) + bc[3])
    wgt = E.softmax(y @ bc[5][:128] + bc[6][0], axis=1)        # logit from conv3's input (u = W3^T wq), as the kernels compute it
    xbar = (wgt[:, :, None] * y).sum(axis=1)
    wa, wb, l2w, l3w = E._split(w['tl_w'], [65536, 32768, 65536, 8192])
    bt = E._split(w['tl_b'].astype(np.float64), [256, 256, 32])
    hh = E.relu(pooled @ E.unpack_dense(wa, 256, 256).T + xbar @ E.unpack_dense(wb, 256, 128).T + bt[0])
    hh = E.relu(hh @ E.unpack_dense(l2w, 256, 256).T + bt[1])
    return hh @ E.unpack_dense(l3w, 2, 256).T + bt[2][:2]

g = load_golden('ppsurf_from_latent')
plan = DecoderPlan(filled_sd('', key='ppsurf'), 'cpu')
w = {k: v.numpy() for k, v in plan.w.items()}
cloud = g['cloud']
for scale in (1.0, 25.0):
    lat = make_latents(256, cloud.shape[0], 77)[0]*scale
    args = (w, lat, cloud, g['query'], g['proj_ids'][0], g['patches'])
    MODE=None; ref = decode(*args)
    print('latent scale', scale, 'logit |max|', np.abs(ref).max())
    for mode in (('f16', 3), ('f16rtz', 3), ('f16', 4), ('bf16', 3), ('bf16', 4), ('f16', 1), ('bf16', 1)):
        MODE=mode; stats.clear()
        out = decode(*args)
        print('  ', mode, 'max logit err {:.3e}'.format(np.abs(out-ref).max()), 'act max {:.1f}'.format(stats['xmax']))
