"""GPU tests of the support-point sampling kernel and the batched neighbourhood tables (C ABI) against the torch-op
implementation of the same contract (source/poco_data_loader.py:59-134) and the kNN oracle."""
import random

import numpy as np
import pytest
import torch

from oracle import ppsurf_oracle as O
from ppsurf_amd import spatial
from ppsurf_amd.synthetic import make_cloud

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('n', [10000, 2500, 625, 156, 39, 9, 2])
def test_sampling_contract_every_level(n):
    random.seed(n); torch.manual_seed(n)
    pts = torch.from_numpy(make_cloud(n, seed=n).T.copy()).unsqueeze(0).to(DEV)
    sup, ids = spatial.sampling_quantized(pts, 0.25)
    target = max(1, int(n * 0.25))
    assert tuple(sup.shape) == (1, 3, target) and tuple(ids.shape) == (1, target) and ids.dtype == torch.int64
    i = ids[0].cpu().numpy()
    assert len(set(i.tolist())) == target and i.min() >= 0 and i.max() < n          # exactly `target` unique valid ids
    assert torch.equal(sup[0], pts[0][:, ids[0]])


def test_full_rounds_match_the_torch_reference_of_the_contract():
    """With the SAME rotations, every point taken in a full round (one representative = smallest index per occupied voxel,
    grid anchored at the rotated bbox minimum, voxel halved each round) is identical between the HIP kernel and the
    torch-op loop; only the random truncation of the last round may differ."""
    random.seed(5)
    rots = spatial.draw_rotations()
    rng = np.random.default_rng(3)                       # clustered cloud: the first voxel sizes hold many points per voxel
    cloud = (rng.uniform(-0.5, 0.5, (200, 1, 3)) + rng.normal(0, 0.004, (200, 50, 3))).reshape(-1, 3).astype(np.float32)
    pts = torch.from_numpy(cloud.T.copy()).unsqueeze(0)
    _, ids_k = spatial.sampling_quantized(pts.to(DEV), 0.25, _rotations=rots)
    # torch-op loop on the CPU (n is small enough for the kernel, so force the fallback by running it on CPU tensors)
    torch.manual_seed(0)
    _, ids_t = spatial.sampling_quantized(pts, 0.25, _rotations=rots)
    a, b = set(ids_k[0].cpu().tolist()), set(ids_t[0].tolist())
    # replay the full rounds to know which ids are not subject to the random truncation
    p = torch.from_numpy(cloud)
    alive = torch.arange(10000)
    ext = (p.max(0)[0] - p.min(0)[0]).norm(2).item()
    vox, full, cnt, r = ext / np.sqrt(2500), [], 0, 0
    while True:
        perm = spatial._one_per_voxel(p[alive] @ rots[r].t(), vox)
        if cnt + perm.shape[0] < 2500:
            full += alive[perm].tolist(); cnt += perm.shape[0]
            keep = torch.ones(alive.shape[0], dtype=torch.bool); keep[perm] = False
            alive = alive[keep]; vox /= 2; r += 1
        else:
            last = set(alive[perm].tolist())
            break
    assert r >= 2 and len(full) > 500 and set(full) <= a and set(full) <= b      # several full rounds happened
    assert (a - set(full)) <= last and (b - set(full)) <= last and len(a) == len(b) == 2500


def test_stratification_is_better_than_random():
    random.seed(1)
    cloud = make_cloud(10000, seed=9)
    pts = torch.from_numpy(cloud.T.copy()).unsqueeze(0).to(DEV)
    _, ids = spatial.sampling_quantized(pts, 0.25)
    sel = cloud[ids[0].cpu().numpy()]
    rnd = cloud[np.random.default_rng(0).choice(10000, 2500, replace=False)]

    def nn_dist(x):
        d = O.knn_point_major(x, x, 2, return_d2=True)[1][:, 1]
        return np.sqrt(d)
    assert nn_dist(sel).min() > nn_dist(rnd).min() and nn_dist(sel).std() < nn_dist(rnd).std()      # blue-noise-like coverage


def test_get_fkaconv_ids_tables_are_exact_knn_of_the_levels():
    random.seed(2)
    pts = torch.from_numpy(make_cloud(10000, seed=4).T.copy()).unsqueeze(0).to(DEV)
    d = spatial.get_fkaconv_ids({'pts': pts})
    levels = [pts] + [d['support{}'.format(i)] for i in (1, 2, 3, 4)]
    assert [lv.shape[2] for lv in levels] == [10000, 2500, 625, 156, 39]
    for a in range(5):
        assert torch.equal(d['ids{}{}'.format(a, a)].cpu(), O.knn(levels[a].cpu(), levels[a].cpu(), 16))
        if a < 4:
            assert torch.equal(d['ids{}{}'.format(a, a + 1)].cpu(), O.knn(levels[a].cpu(), levels[a + 1].cpu(), 16))
            assert torch.equal(d['ids{}{}'.format(a + 1, a)].cpu(), O.knn(levels[a + 1].cpu(), levels[a].cpu(), 1))
    # unbatched input (poco_data_loader.py:143-146) and a tiny cloud whose tables clamp k (poco_utils.py:259-260)
    d2 = spatial.get_fkaconv_ids({'pts': pts[0, :, :300]})
    assert tuple(d2['support1'].shape) == (3, 75) and tuple(d2['ids00'].shape) == (300, 16) and tuple(d2['ids44'].shape) == (1, 1)
    assert tuple(d2['ids34'].shape) == (1, 4) and tuple(d2['ids10'].shape) == (300, 1)
