"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/ppsurf_oracle.py for the rules: only tests/, smoke() and bench.py's
cpu_baseline leg may import anything under oracle/).

CPU restatements of the DRIVER pieces either side of the network:

  sampling_quantized   source/poco_data_loader.py:59-134   support-point sampling (voxel-stratified, halving schedule)
  latent_loop          source/poco_model.py:203-236        coverage-balanced subsets, latent averaging
  refine_vertices      source/poco_utils.py:111-168        bisection of the Marching-Cubes edge vertices

PINNED by tests/golden/make_golden_r2.py: the reference's OWN functions are executed in the build container and their
outputs are stored in tests/golden/{sampling,latent_loop,refine}.npz; tests/test_oracle_golden_r2.py checks these
restatements against them.

Third-party arithmetic under the reference's sampling (not in /root/reference; requirements.txt pins `torch-geometric>=2.3`,
`torch-cluster>=1.6.0`) is restated from its published algorithm:
  * torch_geometric.transforms.RandomRotate(180, axis): angle = pi * random.uniform(-180, 180) / 180 from Python's `random`,
    matrix rows for axis 0 / 1 / 2 as in `_axis_rotation`, applied as `pos @ matrix.t()` (one fp32 matmul per axis);
  * torch_geometric.nn.voxel_grid -> torch_cluster.grid_cluster: grid anchored at the minimum of the (rotated, remaining) points,
    cell = trunc((pos - start) / size), cluster id = cx + cy*nx + cz*nx*ny with n = trunc((end - start) / size) + 1;
  * torch_geometric.nn.pool.consecutive.consecutive_cluster: `unique(sorted)` + `scatter_` of arange -> one representative per
    voxel in ascending cluster-id order; on the CPU scatter_ is sequential, so the LAST (largest) point index of a voxel wins.
For the fixture these three are injected as stand-ins under the reference's sampling_quantized, whose own control flow
(voxel size, removal, halving, truncation by torch.randperm) is what is being pinned.
"""
import math
import random

import numpy as np
import torch


# ----------------------------------------------------------------------------------------------------------------
# a3: sampling_quantized
# ----------------------------------------------------------------------------------------------------------------
def axis_rotation(axis: int, angle_rad: float) -> np.ndarray:
    """torch_geometric RandomRotate matrix for `axis` (float32, like torch.tensor(python floats))."""
    s, c = math.sin(angle_rad), math.cos(angle_rad)
    if axis == 0:
        m = [[1, 0, 0], [0, c, s], [0, -s, c]]
    elif axis == 1:
        m = [[c, 0, -s], [0, 1, 0], [s, 0, c]]
    else:
        m = [[c, s, 0], [-s, c, 0], [0, 0, 1]]
    return np.asarray(m, dtype=np.float32)


def draw_round_rotations():
    """The three matrices of one sampling round in the order the reference draws them: x, y, z
    (`rot_z(rot_y(rot_x(data)))`, poco_data_loader.py:103 -- rot_x is evaluated first)."""
    return [axis_rotation(a, math.pi * random.uniform(-180.0, 180.0) / 180.0) for a in (0, 1, 2)]


def _fma32(a, b, c):
    """fl32(a*b + c) for float32 arrays (one rounding): the product of two floats is exact in float64 and adding a float32
    to it cannot leave double precision by more than a half-ulp double-rounding case that needs >= 29 trailing zeros."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def rotate_f32(pos: np.ndarray, m: np.ndarray) -> np.ndarray:
    """pos [n,3] @ m.T in float32 exactly as the fp32 GEMM microkernel evaluates a K=3 dot product:
    fma(p2, m2, fma(p1, m1, p0 * m0))  (checked bit-for-bit against torch's CPU matmul by the fixture)."""
    out = np.empty_like(pos)
    for c in range(3):
        t = pos[:, 0] * m[c, 0]
        t = _fma32(pos[:, 1], np.full_like(t, m[c, 1]), t)
        out[:, c] = _fma32(pos[:, 2], np.full_like(t, m[c, 2]), t)
    return out


def voxel_cluster(pos: np.ndarray, size: np.float32) -> np.ndarray:
    """torch_cluster.grid_cluster with start = pos.min(0), end = pos.max(0): int64 cluster id per point."""
    start, end = pos.min(axis=0), pos.max(axis=0)
    nvox = ((end - start) / size).astype(np.int64) + 1
    cell = ((pos - start) / size).astype(np.int64)                 # float32 division, truncation
    return cell[:, 0] + cell[:, 1] * nvox[0] + cell[:, 2] * nvox[0] * nvox[1]


def consecutive_representatives(cluster: np.ndarray) -> np.ndarray:
    """consecutive_cluster's `perm`: for each occupied voxel in ascending id order the LARGEST point index in it."""
    order = np.lexsort((np.arange(cluster.shape[0]), cluster))      # by cluster, then by index
    last = np.ones(order.shape[0], dtype=bool)
    last[:-1] = cluster[order][1:] != cluster[order][:-1]
    return order[last]


def initial_voxel_size(pts_n3: np.ndarray, target: int) -> np.float32:
    """poco_data_loader.py:85-88: ||max - min||_2 / sqrt(target) in float32."""
    e = (pts_n3.max(axis=0) - pts_n3.min(axis=0)).astype(np.float32)
    return np.float32(np.float32(np.sqrt(np.float32(np.float32(e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]))) / np.float32(math.sqrt(target)))


def sampling_quantized_ids(pts_n3: np.ndarray, target: int, rotations=None, priority=None, return_rounds=False):
    """One cloud [n,3] float32 -> int64 ids [target] in the reference's order (full rounds in voxel-id order, the last round
    as drawn), poco_data_loader.py:93-125.

    rotations: None -> drawn from Python's `random` exactly as the reference does; or a list of per-round [x, y, z] matrix
               triples (test hook shared with the HIP kernel).
    priority:  None -> the last round is truncated with torch.randperm (CPU generator) like the reference (:123);
               uint32 [n] -> "given the permutation": the representatives with the smallest priority (ties: lower index) are
               kept -- the same distribution when the priorities are a random permutation, and an input the HIP kernel can share.
    """
    pts = np.ascontiguousarray(pts_n3, dtype=np.float32)
    ids = np.arange(pts.shape[0], dtype=np.int64)
    vox = initial_voxel_size(pts, target)
    sampled, count, rnd, rounds = [], 0, 0, []
    while True:
        mats = draw_round_rotations() if rotations is None else rotations[rnd]
        rnd += 1
        rot = pts
        for m in mats:
            rot = rotate_f32(rot, np.asarray(m, dtype=np.float32))
        perm = consecutive_representatives(voxel_cluster(rot, vox))
        if count + perm.shape[0] < target:
            sampled.append(ids[perm])
            rounds.append(ids[perm])
            count += perm.shape[0]
            keep = np.ones(ids.shape[0], dtype=bool)
            keep[perm] = False
            pts, ids = pts[keep], ids[keep]
            vox = np.float32(vox / np.float32(2))
        else:
            need = target - count
            rounds.append(ids[perm])
            if priority is None:
                sel = perm[torch.randperm(perm.shape[0])[:need].numpy()]
            else:
                pr = np.asarray(priority)[ids[perm]]
                sel = perm[np.lexsort((ids[perm], pr))[:need]]
            sampled.append(ids[sel])
            break
    out = np.concatenate(sampled)
    return (out, rounds) if return_rounds else out


def sampling_quantized(pts_batch: torch.Tensor, ratio=None, n_support=None, rotations=None, priority=None):
    """poco_data_loader.py:59-134 for [B,3,N] -> (support [B,3,n], ids int64 [B,n])."""
    assert (ratio is None) != (n_support is None)
    b, _, n = pts_batch.shape
    target = max(1, int(n * ratio)) if ratio is not None else n_support
    if target == n:
        return pts_batch, torch.arange(n, dtype=torch.long).unsqueeze(0).expand(b, n)
    if not 0 < target < n:
        raise ValueError('Search Quantized - ratio value error {} should be in ]0,1]'.format(ratio))
    ids = torch.stack([torch.from_numpy(sampling_quantized_ids(pts_batch[i].t().contiguous().numpy(), target, rotations, priority))
                       for i in range(b)])
    return torch.gather(pts_batch, 2, ids.unsqueeze(1).expand(b, 3, target)), ids


# ----------------------------------------------------------------------------------------------------------------
# a8: latent loop of PocoModel.predict_step
# ----------------------------------------------------------------------------------------------------------------
def latent_loop(pts_n3: torch.Tensor, get_latent, latent_size: int, subsample: int, n_iter: int, trace=None):
    """poco_model.py:203-236.  pts [N,3]; get_latent(pts_subset [1,3,m]) -> [1,C,m].  Returns (latents [N,C], counts [N]).
    Random numbers: torch.randperm on the CPU generator for the subset (:213), torch.randperm on pts.device for the top-up
    (:217-219) -- the same generator here, consumed in the same order.  `trace` collects the id tensor of every pass."""
    n = pts_n3.shape[0]
    latent = torch.zeros((n, latent_size), dtype=torch.float)
    counts = torch.zeros((n,), dtype=torch.float)
    for current_value in range(n_iter):
        while counts.min() < current_value + 1:
            valid_ids = torch.argwhere(counts == current_value)[:, 0].long()
            if n >= subsample:
                ids = valid_ids[torch.randperm(valid_ids.shape[0])[:subsample]]
                if ids.shape[0] < subsample:
                    ids = torch.cat([ids, torch.randperm(n)[:subsample - ids.shape[0]]], dim=0)
            else:
                ids = torch.arange(n)
            part = get_latent(pts_n3[ids].t().unsqueeze(0))
            latent[ids] += part[0].t()                       # duplicate ids (top-up): the last occurrence wins, counted once
            counts[ids] += 1
            if trace is not None:
                trace.append(ids.clone())
    return latent / counts.unsqueeze(1), counts


def stub_latent(pts_cf: torch.Tensor, c: int = 8) -> torch.Tensor:
    """Deterministic stand-in for network.get_latent used by the latent-loop fixture: [B,3,m] -> [B,c,m]; depends on the
    point AND on the subset (its mean), like a real encoder pass does."""
    freq = torch.arange(1, c + 1, dtype=torch.float32).view(1, c, 1)
    centre = pts_cf.mean(dim=2, keepdim=True)
    return torch.sin(freq * pts_cf[:, 0:1]) + torch.cos(freq * pts_cf[:, 1:2]) * pts_cf[:, 2:3] + centre.sum(dim=1, keepdim=True)


# ----------------------------------------------------------------------------------------------------------------
# 8f-2: vertex refinement
# ----------------------------------------------------------------------------------------------------------------
def refine_vertices(verts_grid: np.ndarray, volume: np.ndarray, eval_occ, step, bmin_pad, refine_iter: int, num_pts: int = 50000):
    """poco_utils.py:111-168.  verts_grid [V,3]: Marching-Cubes vertices in GRID coordinates (after the first clean-up);
    volume: float64 (R+2)^3 with NaN = unseen; eval_occ(points float32 [q,3]) -> occ [q].  Returns the vertices in model space
    ([V,3], same dtype as verts_grid) with the edge vertices bisected `refine_iter` times."""
    verts = np.array(verts_grid, copy=True)
    if refine_iter <= 0:
        return verts * step + bmin_pad
    dirs = verts - np.floor(verts)
    dirs = (dirs > 0).astype(dirs.dtype)
    mask = np.logical_and(dirs.sum(axis=1) > 0, dirs.sum(axis=1) < 2)
    v = verts[mask]
    dirs = dirs[mask]
    v1 = np.floor(v)
    v2 = v1 + dirs
    v1 = v1.astype(int)
    v2 = v2.astype(int)
    preds1 = volume[v1[:, 0], v1[:, 1], v1[:, 2]]
    preds2 = volume[v2[:, 0], v2[:, 1], v2[:, 2]]
    v1 = v1.astype(np.float32) * step + bmin_pad
    v2 = v2.astype(np.float32) * step + bmin_pad
    mask_tmp = np.logical_and(np.logical_not(np.isnan(preds1)), np.logical_not(np.isnan(preds2)))
    v = v[mask_tmp]
    v1 = v1[mask_tmp]
    v2 = v2[mask_tmp]
    # NOTE: the reference does NOT filter preds1/preds2 by mask_tmp (poco_utils.py:131-137); whenever a vertex with an unseen
    # corner exists its boolean-index assignments below would raise.  Marching Cubes never emits a vertex on an edge with a
    # NaN corner, so mask_tmp is all-True in every reachable case; the restatement keeps the arrays aligned explicitly.
    preds1 = preds1[mask_tmp]
    preds2 = preds2[mask_tmp]
    mask[mask] = mask_tmp
    verts = verts * step + bmin_pad
    v = v * step + bmin_pad
    for _ in range(refine_iter):
        pnts_all = torch.tensor(v, dtype=torch.float)
        preds = np.concatenate([np.asarray(eval_occ(p.numpy())) for p in torch.split(pnts_all, num_pts, dim=0)], axis=0)
        mask1 = (preds * preds1) > 0
        v1[mask1] = v[mask1]
        preds1[mask1] = preds[mask1]
        mask2 = (preds * preds2) > 0
        v2[mask2] = v[mask2]
        preds2[mask2] = preds[mask2]
        v = (v2 + v1) / 2
        verts[mask] = v
    return verts
