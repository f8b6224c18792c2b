"""GPU parity: brute-force kNN and patch normalisation through the C ABI vs the CPU oracle (bit-exact indices)."""
import numpy as np
import pytest
import torch

from golden_util import load_golden
from oracle import ppsurf_oracle as O
from ppsurf_amd import ops
from ppsurf_amd.synthetic import make_cloud, make_band_queries

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _knn_gpu(pts, qry, k, d2=False):
    r = ops.knn_point_major(torch.from_numpy(pts).to(DEV), torch.from_numpy(qry).to(DEV), k, return_d2=d2)
    torch.cuda.synchronize()
    return (r[0].cpu().numpy(), r[1].cpu().numpy()) if d2 else r.cpu().numpy()


def test_knn_golden_fixture():
    g = load_golden('knn')
    for b in range(2):
        p = np.ascontiguousarray(g['pts'][b].T)
        q = np.ascontiguousarray(g['query'][b].T)
        for key, k in (('ids16', 16), ('ids64', 64), ('ids1', 1)):
            assert np.array_equal(_knn_gpu(p, q, k), g[key].reshape(2, 90, k)[b])


@pytest.mark.parametrize('n,m,k', [(64, 5, 64), (65, 33, 7), (1000, 257, 16), (5000, 1000, 50), (20000, 3000, 64), (39, 156, 1), (9, 4, 9)])
def test_knn_bit_exact_vs_oracle(n, m, k):
    rng = np.random.default_rng(n + m + k)
    pts = rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)
    qry = rng.uniform(-0.5, 0.5, (m, 3)).astype(np.float32)
    idx, d2 = _knn_gpu(pts, qry, k, d2=True)
    ref_idx, ref_d2 = O.knn_point_major(pts, qry, k, return_d2=True)
    assert np.array_equal(idx, ref_idx)
    assert np.array_equal(d2, ref_d2)


def test_knn_ties_quantised_grid():
    # points on a coarse lattice: many exactly equal distances -> order must be (d2, index)
    rng = np.random.default_rng(3)
    pts = (rng.integers(0, 6, (4000, 3)) / 8.0).astype(np.float32)
    qry = (rng.integers(0, 6, (300, 3)) / 8.0).astype(np.float32)
    assert np.array_equal(_knn_gpu(pts, qry, 64), O.knn_point_major(pts, qry, 64))


def test_knn_duplicates_and_self_queries():
    rng = np.random.default_rng(4)
    pts = rng.uniform(-0.5, 0.5, (700, 3)).astype(np.float32)
    pts[100:200] = pts[0:100]
    idx = _knn_gpu(pts, pts, 16)
    assert np.array_equal(idx, O.knn_point_major(pts, pts, 16))
    d = np.linalg.norm(pts[idx[:, 0]] - pts, axis=1)
    assert (d == 0).all()


def test_knn_full_size_properties():
    """BASELINE size (100k points, a 50k-query chunk, k=64): size-independent properties + a sampled oracle check."""
    cloud = make_cloud(100_000, seed=42)
    qry = make_band_queries(cloud, 50_000, resolution=257, seed=1)
    idx, d2 = _knn_gpu(cloud, qry, 64, d2=True)
    assert idx.min() >= 0 and idx.max() < cloud.shape[0]
    assert (np.diff(d2, axis=1) >= 0).all()                       # ascending distances
    srt = np.sort(idx, axis=1)
    assert (np.diff(srt, axis=1) > 0).all()                       # no index twice
    rel = cloud[idx] - qry[:, None, :]
    chk = (rel[..., 0] * rel[..., 0] + rel[..., 1] * rel[..., 1]) + rel[..., 2] * rel[..., 2]
    assert np.array_equal(chk.astype(np.float32), d2)             # reported d2 is the d2 of the reported index
    sel = np.random.default_rng(0).choice(qry.shape[0], 400, replace=False)
    ref_idx = O.knn_point_major(cloud, qry[sel], 64)
    assert np.array_equal(idx[sel], ref_idx)
    assert np.array_equal(idx[:, :50], _knn_gpu(cloud, qry, 50))  # the 50-NN is a prefix of the 64-NN


def test_knn_rejects_bad_arguments():
    from ppsurf_amd._lib import PpsError
    pts = torch.zeros((10, 3), device=DEV)
    with pytest.raises(PpsError):
        ops.knn_point_major(pts, pts, 11)          # k > n
    with pytest.raises(PpsError):
        ops.knn_point_major(pts, pts, 0)
    with pytest.raises(PpsError):
        ops.knn_point_major(pts.cpu(), pts.cpu(), 3)   # no CPU fallback
    assert ops.knn_point_major(pts, pts[:0], 3).shape == (0, 3)


@pytest.mark.parametrize('p', [10, 50, 64, 200])
def test_patch_normalize_vs_oracle(p):
    cloud = make_cloud(5000, seed=2)
    qry = make_band_queries(cloud, 333, resolution=65, seed=5)
    kk = min(p, 64)
    ids = O.knn_point_major(cloud, qry, kk)
    if p > 64:
        ids = np.concatenate([ids] * 4, axis=1)[:, :p].copy()
    out = ops.patch_normalize(torch.from_numpy(cloud).to(DEV), torch.from_numpy(qry).to(DEV), torch.from_numpy(ids).to(DEV), p)
    ref = O.normalize_patches(cloud[ids], qry)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-6, atol=1e-7)
    assert abs(np.linalg.norm(out.cpu().numpy(), axis=2).max(axis=1) - 1.0).max() < 1e-6


@pytest.mark.parametrize('n,m,k', [(9, 4, 9), (63, 10, 5), (64, 70, 64), (65, 33, 7), (1000, 257, 16), (20000, 3000, 64), (100_000, 5000, 64)])
def test_blocked_knn_is_bit_identical_to_brute_force(n, m, k):
    rng = np.random.default_rng(n * 7 + m)
    if n >= 20000:
        pts = make_cloud(n, seed=n)
        qry = make_band_queries(pts, m, resolution=257, seed=m)
    else:
        pts = rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)
        qry = rng.uniform(-0.6, 0.6, (m, 3)).astype(np.float32)
    dp, dq = torch.from_numpy(pts).to(DEV), torch.from_numpy(qry).to(DEV)
    blocks = ops.KnnBlocks(dp)
    idx, d2 = blocks.query(dq, k, return_d2=True)
    ref_idx, ref_d2 = ops.knn_point_major(dp, dq, k, return_d2=True)
    assert torch.equal(idx, ref_idx) and torch.equal(d2, ref_d2)
    assert np.array_equal(idx.cpu().numpy(), O.knn_point_major(pts, qry, k))


@pytest.mark.parametrize('n,m,k', [(4097, 500, 64), (4160, 500, 16), (100_000, 4000, 64), (250_000, 2000, 200), (8192, 300, 130)])
def test_blocked_knn_group_level_changes_nothing(n, m, k):
    """The second level of boxes (groups of 64 blocks), the seed bound bisected from the blocks around the best window and the per-batch
    flushes only prune: with and without the group boxes the search returns the brute-force result bit for bit -- also where the last group
    holds a single (partial) block and for queries far outside the cloud."""
    pts = make_cloud(n, seed=n + k)
    qry = np.concatenate([make_band_queries(pts, m - 20, resolution=257, seed=m),
                          np.random.default_rng(k).uniform(-4, 4, (20, 3)).astype(np.float32)])
    dp, dq = torch.from_numpy(pts).to(DEV), torch.from_numpy(qry).to(DEV)
    blocks = ops.KnnBlocks(dp)
    assert blocks.gbox.shape == ((blocks.nb + 63) // 64, 6)
    a_idx, a_d2 = blocks.query(dq, k, return_d2=True)
    blocks.groups = False
    b_idx, b_d2 = blocks.query(dq, k, return_d2=True)
    assert torch.equal(a_idx, b_idx) and torch.equal(a_d2, b_d2)
    ref_idx, ref_d2 = O.knn_point_major(pts, qry, k, return_d2=True)
    assert np.array_equal(a_idx.cpu().numpy(), ref_idx) and np.array_equal(a_d2.cpu().numpy(), ref_d2)


def test_blocked_knn_ties_and_far_queries():
    rng = np.random.default_rng(11)
    pts = (rng.integers(0, 6, (4000, 3)) / 8.0).astype(np.float32)          # lattice: many exact distance ties
    qry = np.concatenate([(rng.integers(0, 6, (200, 3)) / 8.0), rng.uniform(-30, 30, (50, 3))]).astype(np.float32)
    dp, dq = torch.from_numpy(pts).to(DEV), torch.from_numpy(qry).to(DEV)
    idx = ops.KnnBlocks(dp).query(dq, 64)
    assert np.array_equal(idx.cpu().numpy(), O.knn_point_major(pts, qry, 64))


@pytest.mark.parametrize('n,m,k', [(300, 40, 200), (5000, 333, 65), (5000, 333, 100), (20000, 1000, 128), (100_000, 2000, 200), (3000, 100, 256)])
def test_blocked_knn_large_k_vs_oracle(n, m, k):
    """k up to 256 (the 100NN / 200NN patch searches of configs/ppsurf_100nn.yaml, ppsurf_200nn.yaml)."""
    pts = make_cloud(n, seed=k)
    qry = make_band_queries(pts, m, resolution=129, seed=n)
    idx, d2 = ops.KnnBlocks(torch.from_numpy(pts).to(DEV)).query(torch.from_numpy(qry).to(DEV), k, return_d2=True)
    ref_idx, ref_d2 = O.knn_point_major(pts, qry, k, return_d2=True)
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    assert np.array_equal(d2.cpu().numpy(), ref_d2)


def test_the_k_nearest_are_a_prefix_of_the_p_nearest_also_under_ties():
    """decoder.ChunkPipeline makes ONE search with k = P when the patches (P = 100 / 200) and the projection table (k = 64) come from the
    same cloud, and takes the first 64 columns: the (distance, index) order must make that the table of the separate k = 64 search."""
    rng = np.random.default_rng(3)
    lattice = (rng.integers(0, 7, (6000, 3)) / 8.0).astype(np.float32)      # many exact ties and duplicates
    cloud = np.concatenate([make_cloud(20000, seed=5), lattice]).astype(np.float32)
    qry = np.concatenate([make_band_queries(cloud[:20000], 1500, resolution=129, seed=2), lattice[:300]]).astype(np.float32)
    blocks = ops.KnnBlocks(torch.from_numpy(cloud).to(DEV))
    dq = torch.from_numpy(qry).to(DEV)
    i64 = blocks.query(dq, 64)
    for p in (100, 200):
        assert torch.equal(blocks.query(dq, p)[:, :64], i64)
    assert np.array_equal(i64.cpu().numpy(), O.knn_point_major(cloud, qry, 64))


def test_knn_api_routes_large_k():
    from ppsurf_amd.spatial import knn
    pts = make_cloud(2000, seed=3)
    p = torch.from_numpy(pts.T.copy()).unsqueeze(0).to(DEV)
    ids = knn(p, p[:, :, :50], 200)
    assert tuple(ids.shape) == (1, 50, 200)
    assert np.array_equal(ids[0].cpu().numpy(), O.knn_point_major(pts, pts[:50], 200))
