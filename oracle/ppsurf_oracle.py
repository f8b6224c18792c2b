"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement (torch fp32 on CPU / numpy) of the occupancy-query hot path of cg-tuwien/ppsurf.
Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module;
the product package ppsurf_amd/ never does.

Every function cites the reference lines it restates (paths relative to /root/reference).  The
restatement is PINNED: tests/test_oracle_golden.py checks it against tests/golden/*.npz, which
tests/golden/make_golden.py produced by importing and running the reference's own modules in the
build container (formula-filled parameters, ppsurf_amd/synthetic.py).  The only unpinned piece is
kNN tie order (third-party pykdtree, not in the tree) -- see oracle/knn_oracle.c.

Functional style: parameters come from a flat {state_dict_name: tensor} dict `sd` with the
reference's names (e.g. 'projection.fc1.weight'); `p` is the prefix of the sub-module.
Norm layers are evaluated in eval() mode (running statistics) unless stated.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
BN_EPS = 1e-5


# ----------------------------------------------------------------------------------------------
# kNN (source/poco_utils.py:257-273, source/base/proximity.py:40-89)
# ----------------------------------------------------------------------------------------------
_knn_lib = None


def build_c_oracle(force=False):
    """gcc build of oracle/knn_oracle.c -> oracle/libknn_oracle.so (ignored by git, travels with gpurun)."""
    so = os.path.join(_HERE, 'libknn_oracle.so')
    src = os.path.join(_HERE, 'knn_oracle.c')
    if force or not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fno-fast-math', '-fopenmp', '-shared', '-fPIC',
                               src, '-o', so])
    return so


def _lib():
    global _knn_lib
    if _knn_lib is None:
        _knn_lib = ctypes.CDLL(build_c_oracle())
        _knn_lib.pps_oracle_knn_f32.restype = ctypes.c_int
        _knn_lib.pps_oracle_knn_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                                ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return _knn_lib


def knn_point_major(pts: np.ndarray, query: np.ndarray, k: int, return_d2=False):
    """pts [n,3] f32, query [m,3] f32 -> int64 [m,k] sorted by (d2, index); d2 = (dx*dx+dy*dy)+dz*dz in f32."""
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    query = np.ascontiguousarray(query, dtype=np.float32)
    n, m = pts.shape[0], query.shape[0]
    idx = np.empty((m, k), dtype=np.int64)
    d2 = np.empty((m, k), dtype=np.float32)
    if m > 0:
        rc = _lib().pps_oracle_knn_f32(pts.ctypes.data, n, query.ctypes.data, m, int(k), idx.ctypes.data, d2.ctypes.data)
        if rc != 0:
            raise ValueError('knn oracle: bad k={} for n={}'.format(k, n))
    return (idx, d2) if return_d2 else idx


def knn_numpy(pts: np.ndarray, query: np.ndarray, k: int):
    """Pure-numpy twin of knn_point_major for small cases (cross-checks the C build)."""
    pts = pts.astype(np.float32)
    query = query.astype(np.float32)
    d = query[:, None, :] - pts[None, :, :]
    xx = d[..., 0] * d[..., 0]
    yy = d[..., 1] * d[..., 1]
    zz = d[..., 2] * d[..., 2]
    d2 = (xx + yy) + zz
    order = np.lexsort((np.broadcast_to(np.arange(pts.shape[0]), d2.shape), d2), axis=1)
    return order[:, :k].astype(np.int64)


def knn(points: torch.Tensor, support_points: torch.Tensor, k: int) -> torch.Tensor:
    """source/poco_utils.py:257-273: points [B,3,N], support [B,3,M] -> int64 [B,M,k]; k clamped to N."""
    k = min(int(k), points.shape[2])
    out = []
    for b in range(points.shape[0]):
        p = points[b].detach().cpu().transpose(0, 1).contiguous().numpy()
        s = support_points[b].detach().cpu().transpose(0, 1).contiguous().numpy()
        out.append(torch.from_numpy(knn_point_major(p, s, k)))
    return torch.stack(out, dim=0)


# ----------------------------------------------------------------------------------------------
# gathers (source/base/nn.py:655-697)
# ----------------------------------------------------------------------------------------------
def batch_gather(data: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """nn.py:655-674 for dim=2: data [B,C,N], index [B,M,K] -> [B,C,M,K]."""
    b, c, _ = data.shape
    _, m, k = index.shape
    flat = index.reshape(b, 1, m * k).expand(b, c, m * k)
    return torch.gather(data, 2, flat).reshape(b, c, m, k)


def max_pool(data, index):
    """nn.py:677-680."""
    return batch_gather(data, index).max(dim=3)[0]


def interpolate(x, index):
    """nn.py:684-697 (method='mean'); negative ids are clamped to 0 without mutating the input."""
    idx = torch.where(index > -1, index, torch.zeros_like(index))
    g = batch_gather(x, idx)
    return g.mean(-1) if idx.shape[-1] > 1 else g.squeeze(-1)


# ----------------------------------------------------------------------------------------------
# small layer helpers
# ----------------------------------------------------------------------------------------------
def _bn(sd, p, x, channel_dim=1):
    """BatchNorm eval: (x-rm)/sqrt(rv+eps)*w+b."""
    shape = [1] * x.dim()
    shape[channel_dim] = -1
    rm, rv = sd[p + '.running_mean'].view(shape), sd[p + '.running_var'].view(shape)
    w, b = sd[p + '.weight'].view(shape), sd[p + '.bias'].view(shape)
    return (x - rm) / torch.sqrt(rv + BN_EPS) * w + b


def _conv1(sd, p, x, bias=True):
    """1x1 Conv1d / Conv2d on channel dim 1 == per-position matmul."""
    w = sd[p + '.weight']
    w = w.reshape(w.shape[0], w.shape[1])
    y = torch.einsum('oc,bc...->bo...', w, x)
    if bias:
        shape = [1, -1] + [1] * (x.dim() - 2)
        y = y + sd[p + '.bias'].view(shape)
    return y


def _linear(sd, p, x):
    return x @ sd[p + '.weight'].t() + sd[p + '.bias']


def _instance_norm(sd, p, x):
    """InstanceNorm2d(affine, eps 1e-5, biased var, no running stats): stats over dims (2,3) (nn.py:586-587)."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    y = (x - mean) / torch.sqrt(var + 1e-5)
    return y * sd[p + '.weight'].view(1, -1, 1, 1) + sd[p + '.bias'].view(1, -1, 1, 1)


def _act(name):
    return {'relu': torch.relu, 'silu': torch.nn.functional.silu}[name]


# ----------------------------------------------------------------------------------------------
# FKAConv encoder (source/base/nn.py:420-652)
# ----------------------------------------------------------------------------------------------
def fkaconv_layer(sd, p, x, pts, support, ids, act='relu'):
    """nn.py:592-652 (eval mode: norm_radius is not updated).  x [B,Cin,N], pts [B,3,N], support [B,3,Ns],
    ids [B,Ns,K] -> [B,Cout,Ns]."""
    f = _act(act)
    k = ids.shape[2]
    pn = batch_gather(pts, ids) - support.unsqueeze(3)                      # :597,601
    xg = batch_gather(x, ids)                                               # :598
    dist = torch.sqrt((pn ** 2).sum(1))                                     # :605  [B,Ns,K]
    pn = pn / sd[p + '.norm_radius']                                        # :616
    dw = torch.sigmoid(-sd[p + '.alpha'] * dist + sd[p + '.beta'])          # :619
    s = dw.sum(2, keepdim=True)
    s = s + (s == 0) + 1e-6                                                 # :620-621
    dw = (dw / s * k).unsqueeze(1)                                          # :622-624  [B,1,Ns,K]
    m = _conv1(sd, p + '.fc1', pn, bias=False)
    m = f(m if k == 1 else _instance_norm(sd, p + '.bn1', m))               # :627-630
    mp = (m * dw).max(dim=3, keepdim=True)[0].expand(-1, -1, -1, k)         # :631-633
    m = _conv1(sd, p + '.fc2', torch.cat([m, mp], dim=1), bias=False)
    m = f(m if k == 1 else _instance_norm(sd, p + '.bn2', m))               # :635-638
    mp = (m * dw).max(dim=3, keepdim=True)[0].expand(-1, -1, -1, k)         # :639-641
    m = f(_conv1(sd, p + '.fc3', torch.cat([m, mp], dim=1), bias=False)) * dw   # :643
    feat = torch.einsum('bcmk,btmk->bcmt', xg, m)                           # :647-649  [B,Cin,Ns,16]
    w = sd[p + '.cv.weight'][:, :, 0, :]                                    # [Cout,Cin,16]
    return torch.einsum('oct,bcmt->bom', w, feat)                           # :650


def residual_block(sd, p, x, pts, support, ids, act='relu'):
    """nn.py:438-450."""
    h = torch.relu(_bn(sd, p + '.bn0', _conv1(sd, p + '.cv0', x)))
    h = torch.relu(_bn(sd, p + '.bn1', fkaconv_layer(sd, p + '.cv1', h, pts, support, ids, act)))
    h = _bn(sd, p + '.bn2', _conv1(sd, p + '.cv2', h))
    sc = x
    if (p + '.shortcut.weight') in sd:
        sc = _bn(sd, p + '.bn_shortcut', _conv1(sd, p + '.shortcut', sc))
    if sc.shape[2] != h.shape[2]:
        sc = max_pool(sc, ids)
    return torch.relu(h + sc)


def fkaconv_network(sd, p, data, act='relu', fixed=False):
    """nn.py:508-554 with spectral_only=True, segmentation=True, dropout p=0.  data holds pts, support1..4, ids*."""
    pts = data['pts']
    x = torch.ones_like(pts)
    s1, s2, s3, s4 = data['support1'], data['support2'], data['support3'], data['support4']
    x0 = torch.relu(_bn(sd, p + '.bn0', fkaconv_layer(sd, p + '.cv0', x, pts, pts, data['ids00'], act)))
    x0 = residual_block(sd, p + '.resnetb01', x0, pts, pts, data['ids00'], act)
    x1 = residual_block(sd, p + '.resnetb10', x0, pts, s1, data['ids01'], act)
    x1 = residual_block(sd, p + '.resnetb11', x1, s1, s1, data['ids11'], act)
    x2 = residual_block(sd, p + '.resnetb20', x1, s1, s2, data['ids12'], act)
    x2 = residual_block(sd, p + '.resnetb21', x2, s2, s2, data['ids22'], act)
    x3 = residual_block(sd, p + '.resnetb30', x2, s2, s3, data['ids23'], act)
    x3 = residual_block(sd, p + '.resnetb31', x3, s3, s3, data['ids33'], act)
    x4 = residual_block(sd, p + '.resnetb40', x3, s3, s4, data['ids34'], act)
    x4 = residual_block(sd, p + '.resnetb41', x4, s4, s4, data['ids44'], act)
    x5 = x4.max(dim=2, keepdim=True)[0].expand_as(x4)                                     # :531
    x4d = torch.relu(_bn(sd, p + '.bn5', _conv1(sd, p + '.cv5', torch.cat([x4, x5], 1))))  # :532
    if not fixed:
        x4d = x4                                                                          # :533-534
    x3d = torch.relu(_bn(sd, p + '.bn3d', _conv1(sd, p + '.cv3d', torch.cat([interpolate(x4d, data['ids43']), x3], 1))))
    x2d = torch.relu(_bn(sd, p + '.bn2d', _conv1(sd, p + '.cv2d', torch.cat([interpolate(x3d, data['ids32']), x2], 1))))
    x1d = torch.relu(_bn(sd, p + '.bn1d', _conv1(sd, p + '.cv1d', torch.cat([interpolate(x2d, data['ids21']), x1], 1))))
    xo = torch.relu(_bn(sd, p + '.bn0d', _conv1(sd, p + '.cv0d', torch.cat([interpolate(x1d, data['ids10']), x0], 1))))
    return _conv1(sd, p + '.fcout', xo)                                                   # :547-548


def fkaconv_ids_from_supports(pts, supports):
    """The 13 id tables of source/poco_data_loader.py:155-168 for GIVEN support levels (sampling bypassed).
    pts [B,3,N0]; supports = [support1..4] -> dict."""
    s = [pts] + list(supports)
    out = {'support1': s[1], 'support2': s[2], 'support3': s[3], 'support4': s[4]}
    for a in range(5):
        out['ids{}{}'.format(a, a)] = knn(s[a], s[a], 16)
        if a < 4:
            out['ids{}{}'.format(a, a + 1)] = knn(s[a], s[a + 1], 16)
            out['ids{}{}'.format(a + 1, a)] = knn(s[a + 1], s[a], 1)
    return out


# ----------------------------------------------------------------------------------------------
# decoder (source/poco_model.py:362-419, source/base/nn.py:72-96,133-190,255-417, source/ppsurf_model.py:82-117)
# ----------------------------------------------------------------------------------------------
def interp_attention(sd, p, latents, proj_ids, pts, pts_query, last_layer=True):
    """poco_model.py:381-419.  latents [B,C,N], proj_ids [B,Q,k], pts [B,3,N], pts_query [B,3,Q] -> [B,Cout,Q]."""
    x = batch_gather(latents, proj_ids)                                    # :400
    rel = pts_query.unsqueeze(3) - batch_gather(pts, proj_ids)             # :401-402  query minus neighbour
    x = torch.cat([x, rel], dim=1)                                         # :404
    x = torch.relu(_conv1(sd, p + '.fc1', x))
    x = torch.relu(_conv1(sd, p + '.fc2', x))
    x = torch.relu(_conv1(sd, p + '.fc3', x))                              # :405-407
    query = _conv1(sd, p + '.fc_query', x)                                 # [B,64,Q,k]
    value = _conv1(sd, p + '.fc_value', x)                                 # [B,C,Q,k]
    att = torch.softmax(query, dim=-1).mean(dim=1)                         # :412  [B,Q,k]
    x = torch.einsum('bqk,bcqk->bcq', att, value)                          # :413-414
    if last_layer:
        x = _conv1(sd, p + '.fc8', x)                                      # :416-417
    return x


def stn(sd, p, x):
    """nn.py:162-190 (num_scales=1, dim = x.shape[1]).  x [Q,D,P] -> [Q,D,D]."""
    d = x.shape[1]
    h = torch.relu(_bn(sd, p + '.bn1', _conv1(sd, p + '.conv1', x)))
    h = torch.relu(_bn(sd, p + '.bn2', _conv1(sd, p + '.conv2', h)))
    h = torch.relu(_bn(sd, p + '.bn3', _conv1(sd, p + '.conv3', h)))
    h = h.max(dim=2)[0]                                                    # MaxPool1d(num_points) :170
    h = torch.relu(_bn(sd, p + '.bn4', _linear(sd, p + '.fc1', h)))
    h = torch.relu(_bn(sd, p + '.bn5', _linear(sd, p + '.fc2', h)))
    h = _linear(sd, p + '.fc3', h)
    return (h + torch.eye(d, dtype=h.dtype).reshape(1, d * d)).view(-1, d, d)


def attention_poco(sd, p, x):
    """nn.py:84-96 (reduce=True).  x [Q,C,P] -> [Q,C]."""
    w = torch.softmax(_conv1(sd, p + '.fc_query', x)[:, 0, :], dim=-1)     # [Q,P]
    v = _conv1(sd, p + '.fc_value', x)                                     # [Q,C,P]
    return torch.einsum('qp,qcp->qc', w, v)


def pointnet_feat(sd, p, x):
    """nn.py:305-373 with use_point_stn=False, use_feat_stn=True, sym_op='att', num_scales=1.
    x [Q,3,P] -> (feat [Q,out], trans2 [Q,64,64])."""
    h = torch.relu(_bn(sd, p + '.bn0a', _conv1(sd, p + '.conv0a', x)))
    h = torch.relu(_bn(sd, p + '.bn0b', _conv1(sd, p + '.conv0b', h)))     # :323-324
    trans2 = stn(sd, p + '.stn2', h)
    h = torch.bmm(trans2, h)                                               # :327-329
    h = torch.relu(_bn(sd, p + '.bn1', _conv1(sd, p + '.conv1', h)))
    h = torch.relu(_bn(sd, p + '.bn2', _conv1(sd, p + '.conv2', h)))
    h = _bn(sd, p + '.bn3', _conv1(sd, p + '.conv3', h))                   # :334-336 (no ReLU)
    return attention_poco(sd, p + '.att', h), trans2


def mlp(sd, p, x, num_layers=3):
    """nn.py:376-417 (halving_size=False, eval: dropout off).  x [Q,C] -> [Q,out]."""
    for i in range(num_layers - 1):
        x = torch.relu(_bn(sd, '{}.layers.{}.1'.format(p, i), _linear(sd, '{}.layers.{}.0'.format(p, i), x)))
    return _linear(sd, '{}.layers.{}.0'.format(p, num_layers - 1), x)


def _channel_first(t):
    return t if t.shape[1] == 3 else t.transpose(1, 2)


def ppsurf_from_latent(sd, data, k=64, p=''):
    """source/ppsurf_model.py:82-117.  data: latents [B,C,N], pts [B,3,N], pts_query [B,Q,3]|[B,3,Q],
    pts_local_ps [B,Q,P,3] -> logits [B,2,Q].  proj_ids are always recomputed (has_proj_ids=False, :83)."""
    pts = _channel_first(data['pts'])
    ptq = _channel_first(data['pts_query'])
    proj_ids = knn(pts, ptq, k)                                            # poco_data_loader.py:212-240
    feat_proj = interp_attention(sd, p + 'projection', data['latents'], proj_ids, pts, ptq)
    pl = data['pts_local_ps']
    b, q = pl.shape[0], pl.shape[1]
    feat_pn, _ = pointnet_feat(sd, p + 'point_net', pl.reshape(b * q, pl.shape[2], 3).transpose(1, 2))
    feat = feat_proj.transpose(1, 2) + feat_pn.view(b, q, -1)              # :100 (stack+sum)
    out = mlp(sd, p + 'mlp', feat.reshape(b * q, -1))
    return out.view(b, q, -1).transpose(1, 2)


def poco_forward_from_latent(sd, data, p=''):
    """source/poco_model.py:345-359 `PocoNetwork.forward` tail: projection with PRECOMPUTED proj_ids."""
    pts = _channel_first(data['pts'])
    ptq = _channel_first(data['pts_query'])
    return interp_attention(sd, p + 'projection', data['latents'], data['proj_ids'], pts, ptq)


def predict_from_latent(logits):
    """source/poco_utils.py:74-82: softmax over the 2 classes, occ = p0 - p1.  logits [1,2,q] -> [q]."""
    pr = torch.softmax(logits, dim=1)
    return (pr[:, 0] - pr[:, 1]).squeeze(0)


# ----------------------------------------------------------------------------------------------
# patches (source/ppsurf_data_loader.py:83-123, source/poco_utils.py:67-72)
# ----------------------------------------------------------------------------------------------
def normalize_patches(pts_local_ms: np.ndarray, pts_query_ms: np.ndarray) -> np.ndarray:
    """ppsurf_data_loader.py:91-123: centre at the query, divide by max neighbour distance.  [Q,P,3],[Q,3] -> [Q,P,3]."""
    rel = pts_local_ms - pts_query_ms[:, None, :]
    radius = np.sqrt((rel * rel).sum(axis=2)).max(axis=1)                  # np.linalg.norm(.., axis=2) then max :107-109
    return rel / radius[:, None, None]


def get_pts_local_ps(pts_raw_ms: np.ndarray, pts_query: np.ndarray, num_pts_local: int) -> np.ndarray:
    """poco_utils.py:67-72: k=P NN of each query in the raw cloud, gather, normalise -> [Q,P,3]."""
    ids = knn_point_major(pts_raw_ms, pts_query, num_pts_local)
    return normalize_patches(pts_raw_ms[ids], pts_query.astype(np.float32))


# ----------------------------------------------------------------------------------------------
# loss / metrics shell (source/poco_model.py:75-101, source/base/metrics.py:41-84, poco_data_loader.py:251-255)
# ----------------------------------------------------------------------------------------------
def occ_labels(imp_surf_dist_ms: torch.Tensor) -> torch.Tensor:
    return (torch.sign(imp_surf_dist_ms) > 0.0).to(torch.int64)


def compute_loss(pred, occ):
    return torch.nn.functional.cross_entropy(pred, occ, reduction='none').mean()


def binary_metrics(pred_logits, occ):
    lab = torch.argmax(pred_logits, dim=1).squeeze() > 0
    gt = occ.squeeze() > 0
    tp = float((lab & gt).sum()); fp = float((lab & ~gt).sum()); fn = float((~lab & gt).sum()); tn = float((~lab & ~gt).sum())
    n = tp + fp + fn + tn
    nan = float('nan')
    prec = tp / (tp + fp) if tp + fp > 0 else nan
    rec = tp / (tp + fn) if tp + fn > 0 else nan
    f1 = 2.0 * prec * rec / (prec + rec) if (prec == prec and rec == rec and prec + rec > 0) else nan
    return {'accuracy': (tp + tn) / n if n > 0 else nan, 'precision': prec, 'recall': rec, 'f1_score': f1,
            'true_pos': tp, 'false_pos': fp, 'false_neg': fn, 'true_neg': tn}


# ----------------------------------------------------------------------------------------------
# region-growing volume (source/poco_utils.py:178-254) -- "next" row 8(f)-1
# ----------------------------------------------------------------------------------------------
def create_volume(eval_occ, pts_ids, resolution, step, bmin_pad, num_pts, padding=1, dilation_size=2, out_value=1.0):
    """poco_utils.py:178-254 with the network replaced by eval_occ(points[q,3] f32) -> occ[q].
    Returns (volume float64 (R+2p)^3 with NaN = unseen, number of evaluated queries)."""
    shape = (resolution + 2 * padding,) * 3
    volume = np.full(shape, np.nan, dtype=np.float64)
    to_see = np.ones(shape, dtype=bool)
    n_eval = 0

    def dilate(arr, ids):
        lo = np.maximum(0, ids - dilation_size)
        hi = np.minimum(arr.shape[0], ids + dilation_size + 1)
        for a, b in zip(lo, hi):
            arr[a[0]:b[0], a[1]:b[1], a[2]:b[2]] = True
        return arr

    while pts_ids.shape[0] > 0:
        mask = np.zeros(shape, dtype=bool)
        mask[pts_ids[:, 0], pts_ids[:, 1], pts_ids[:, 2]] = True
        mask = dilate(mask, pts_ids)
        valid = np.argwhere(mask).astype(np.float32) * step + bmin_pad
        z = [np.asarray(eval_occ(valid[s:s + num_pts].astype(np.float32))) for s in range(0, valid.shape[0], num_pts)]
        n_eval += valid.shape[0]
        volume[mask] = np.concatenate(z, axis=0).astype(np.float64)
        to_see[pts_ids[:, 0], pts_ids[:, 1], pts_ids[:, 2]] = False
        v = volume[pts_ids[:, 0], pts_ids[:, 1], pts_ids[:, 2]]
        mask_neg = dilate(np.zeros(shape, dtype=bool), pts_ids[v <= 0])
        mask_pos = dilate(np.zeros(shape, dtype=bool), pts_ids[v >= 0])
        with np.errstate(invalid='ignore'):
            new_mask = (mask_neg & (volume >= 0) & to_see) | (mask_pos & (volume <= 0) & to_see)
        pts_ids = np.argwhere(new_mask).astype(np.int64)
    pad = padding
    volume[0:pad] = out_value; volume[-pad:] = out_value
    volume[:, 0:pad] = out_value; volume[:, -pad:] = out_value
    volume[:, :, 0:pad] = out_value; volume[:, :, -pad:] = out_value
    return volume, n_eval
