"""Per-kernel timing of the occupancy-query path on one GPU (development aid; bench.py is the judged entry point)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from golden_util import filled_sd
from ppsurf_amd import ops
from ppsurf_amd.decoder import DecoderPlan
from ppsurf_amd.synthetic import make_cloud, make_band_queries, make_latents

DEV = 'cuda:0'
N = int(os.environ.get('N', 100000)); Q = int(os.environ.get('Q', 50000)); REP = int(os.environ.get('REP', 5))
pl = DecoderPlan(filled_sd('', key='ppsurf'), DEV)
cloud = make_cloud(N, seed=42)
qry = make_band_queries(cloud, Q, resolution=257, seed=1)
pts, qd = torch.from_numpy(cloud).to(DEV), torch.from_numpy(qry).to(DEV)
lat = torch.from_numpy(make_latents(256, N, 77)[0]).to(DEV)


def timeit(name, fn, rep=REP):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(rep + 1)]
    ev[0].record()
    for i in range(rep):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(rep)]
    print('{:28s} min {:9.3f} ms  med {:9.3f} ms'.format(name, min(ts), sorted(ts)[len(ts) // 2]), flush=True)
    return min(ts)


t_tab = timeit('point_table (N rows)', lambda: pl.point_table(lat))
table = pl.point_table(lat)
t_knn = timeit('knn k=64', lambda: ops.knn_point_major(pts, qd, 64))
idx = ops.knn_point_major(pts, qd, 64)
blocks = ops.KnnBlocks(pts)
t_kb = timeit('knn blocked k=64', lambda: blocks.query(qd, 64))
t_bld = timeit('KnnBlocks build', lambda: ops.KnnBlocks(pts))
t_pat = timeit('patch_normalize P=50', lambda: ops.patch_normalize(pts, qd, idx, 50))
patches = ops.patch_normalize(pts, qd, idx, 50)
from ppsurf_amd import _lib
L = _lib.lib(); w = pl.w; st = torch.cuda.current_stream().cuda_stream
pooled = pl.scratch('pooled', (Q, 256)); g = pl.scratch('g', (Q, 256)); tr = pl.scratch('trans2', (Q, 4096)); xb = pl.scratch('xbar', (Q, 256))
logits = torch.empty((Q, 2), device=DEV); occ = torch.empty((Q,), device=DEV)
t_ip = timeit('interp_pool', lambda: L.pps_interp_pool_f32(table.data_ptr(), pts.data_ptr(), qd.data_ptr(), idx.data_ptr(), Q, 64, w['ip_w'].data_ptr(), w['ip_b'].data_ptr(), pooled.data_ptr(), st))
t_pa = timeit('pointnet_stn_rows', lambda: L.pps_pointnet_stn_rows_f32(patches.data_ptr(), Q, 50, w['pa_w'].data_ptr(), w['pa_b'].data_ptr(), g.data_ptr(), st))
t_pb = timeit('pointnet_stn_fc', lambda: L.pps_pointnet_stn_fc_f32(g.data_ptr(), Q, w['pb_w'].data_ptr(), w['pb_b'].data_ptr(), tr.data_ptr(), st))
t_pc = timeit('pointnet_feat_rows', lambda: L.pps_pointnet_feat_rows_f32(patches.data_ptr(), tr.data_ptr(), Q, 50, w['pc_w'].data_ptr(), w['pc_b'].data_ptr(), xb.data_ptr(), st))
t_tl = timeit('decode_tail', lambda: L.pps_decode_tail_f32(pooled.data_ptr(), xb.data_ptr(), Q, w['tl_w'].data_ptr(), w['tl_b'].data_ptr(), logits.data_ptr(), occ.data_ptr(), st))
t_all = timeit('knn+patch+decode (chunk)', lambda: pl.decode(table, pts, qd, ops.knn_point_major(pts, qd, 64), ops.patch_normalize(pts, qd, idx, 50)))
mf = {'interp_pool': 2320 * 4, 'pointnet_stn_rows': 772 * 4, 'pointnet_stn_fc': 4736 / 16, 'pointnet_feat_rows': 1040, 'decode_tail': 2688 / 16}
for name, t in (('interp_pool', t_ip), ('pointnet_stn_rows', t_pa), ('pointnet_stn_fc', t_pb), ('pointnet_feat_rows', t_pc), ('decode_tail', t_tl)):
    fl = mf[name] * 2048.0 * Q
    print('{:20s} executed {:7.1f} TFLOP/s ({:.1%} of 157.3)'.format(name, fl / t / 1e9, fl / t / 1e9 / 157.3))
print('queries/s (chunk incl. kNN): {:.3e}'.format(Q / t_all * 1e3))
print('algorithmic TFLOP/s at 53.21 MFLOP/query: {:.1f}'.format(Q * 53.21e6 / (t_all - t_knn - t_pat) / 1e9))
