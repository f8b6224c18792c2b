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
