"""Support sampling + the 130 id tables of a batch of 10 x 10k-point clouds (what every fit step and every latent-loop batch does)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppsurf_amd import spatial
from ppsurf_amd.synthetic import make_cloud
DEV = 'cuda:0'
pts = torch.from_numpy(np.stack([make_cloud(10000, seed=i) for i in range(10)])).to(DEV).transpose(1, 2).contiguous()
for thr in (10**9, 1024):
    spatial.BLOCKED_MIN_POINTS = thr
    for _ in range(3):
        spatial.get_fkaconv_ids({'pts': pts})
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        spatial.get_fkaconv_ids({'pts': pts})
    torch.cuda.synchronize()
    print('blocked tables for levels >= {:>10d} points: {:.2f} ms per batch (sampling + 130 tables)'.format(thr, (time.perf_counter() - t0) / 20 * 1e3))

from ppsurf_amd import ops
def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
lv = [pts.transpose(1, 2).contiguous()]
for _ in range(4):
    cur = lv[-1]; lv.append(cur[:, ::4].contiguous())
t0_, b0 = timed(lambda: ops.BlockedLevel(lv[0])); t1_, b1 = timed(lambda: ops.BlockedLevel(lv[1]))
print('prep level0 {:.3f} ms, level1 {:.3f} ms'.format(t0_, t1_))
kinds = [(b0, b0, 16), (b0, b1, 16), (b1, b0, 1), (b1, b1, 16), (b1, lv[2], 16)]
for i, kd in enumerate(kinds):
    print('kind', i, '{:.3f} ms'.format(timed(lambda: ops.knn_blocked_batch([kd]))[0]))
print('all 5 kinds one launch {:.3f} ms'.format(timed(lambda: ops.knn_blocked_batch(kinds))[0]))
ps, qs, ks = [], [], []
for pa, qa, k in ((0, 0, 16), (0, 1, 16), (1, 0, 1), (1, 1, 16), (1, 2, 16)):
    for b in range(10):
        ps.append(lv[pa][b]); qs.append(lv[qa][b]); ks.append(k)
print('same 5 tables exhaustive {:.3f} ms'.format(timed(lambda: ops.knn_batch_point_major(ps, qs, ks))[0]))
print('sampling 4 levels {:.3f} ms'.format(timed(lambda: [spatial.voxel_sample_batch_point_major(l, l.shape[1] // 4) for l in lv[:4]])[0]))
