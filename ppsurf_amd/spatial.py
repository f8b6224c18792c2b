"""Spatial queries and neighbourhood assembly on the GPU, behind the reference's free-function API.

  knn                -> source/poco_utils.py:257-273            (CPU kd-tree per call in the reference)
  sampling_quantized -> source/poco_data_loader.py:59-134      (torch_geometric voxel_grid + consecutive_cluster)
  get_fkaconv_ids    -> source/poco_data_loader.py:137-209
  get_proj_ids       -> source/poco_data_loader.py:212-240
  get_data_poco      -> source/poco_data_loader.py:243-270
  normalize_patches / get_pts_local_ps -> source/ppsurf_data_loader.py:91-123, source/poco_utils.py:67-72

Tensors keep the reference's channel-first [B,3,N] layout at this boundary; the kernels work point-major.
"""
import ctypes
import math
import random

import torch

from . import ops, _lib

N_ROT = 12          # rotations drawn per sampling level (rounds beyond that would need 2^-12 of the initial voxel edge)


def _point_major(t):
    """[3,N] (or [N,3] view) -> contiguous [N,3] float32."""
    return t.transpose(0, 1).contiguous().float()


def knn(points: torch.Tensor, support_points: torch.Tensor, k: int, workers: int = 1) -> torch.Tensor:
    """points [B,3,N], support_points [B,3,M] -> int64 [B,M,k] on the device of `points`; k clamps to N
    (poco_utils.py:259-260).  `workers` is accepted and ignored, as the reference ignores it for pykdtree."""
    k = min(int(k), points.shape[2])
    nb = points.shape[0]
    if k <= 64 and nb > 1 and points.is_cuda:          # a fit batch: all shapes in one launch
        out = ops.knn_batch_point_major([_point_major(points[b]) for b in range(nb)], [_point_major(support_points[b]) for b in range(nb)],
                                        [k] * nb)
    elif k <= 64:
        out = [ops.knn_point_major(_point_major(points[b]), _point_major(support_points[b]), k) for b in range(points.shape[0])]
    else:                                   # 64 < k <= 256 (100NN / 200NN patches): block-culling search, same results
        out = [ops.KnnBlocks(_point_major(points[b])).query(_point_major(support_points[b]), k) for b in range(points.shape[0])]
    return torch.stack(out, dim=0)


def _axis_rotation(axis: int, angle: float):
    """torch_geometric RandomRotate matrix about one axis (applied as pos @ m.t())."""
    s, c = math.sin(angle), math.cos(angle)
    if axis == 0:
        return [[1, 0, 0], [0, c, s], [0, -s, c]]
    if axis == 1:
        return [[c, 0, -s], [0, 1, 0], [s, 0, c]]
    return [[c, s, 0], [-s, c, 0], [0, 0, 1]]


def draw_rotations(n: int = N_ROT, batch: int = None) -> torch.Tensor:
    """The axis rotations of n sampling rounds: per round three matrices in the order the reference draws and applies them
    (x, y, z; angles U(-180,180) degrees from Python's `random` like torch_geometric's RandomRotate, poco_data_loader.py:90-103)
    -> float32 [n,3,3,3] ([batch,n,3,3,3]) on the CPU.  The reference draws one triple per round it actually runs; drawing N_ROT
    of them up front lets all rounds run in one launch (the extra draws only advance Python's random stream)."""
    total = n * (batch or 1)
    r = torch.tensor([[_axis_rotation(a, math.pi * random.uniform(-180.0, 180.0) / 180.0) for a in (0, 1, 2)] for _ in range(total)],
                     dtype=torch.float32)
    return r.view(batch, n, 3, 3, 3) if batch else r


def _representatives(pos: torch.Tensor, size) -> torch.Tensor:
    """One representative per occupied voxel of edge `size`: torch_geometric's voxel_grid (grid anchored at the minimum, cell =
    trunc((pos - start) / size), id = cx + cy*nx + cz*nx*ny) followed by consecutive_cluster's `perm` -- voxels in ascending
    id order, the LARGEST point index of each voxel (what the sequential CPU scatter_ leaves; oracle/driver_oracle.py)."""
    start, end = pos.min(dim=0)[0], pos.max(dim=0)[0]
    nvox = ((end - start) / size).to(torch.long) + 1
    cell = ((pos - start) / size).to(torch.long)
    key = cell[:, 0] + cell[:, 1] * nvox[0] + cell[:, 2] * nvox[0] * nvox[1]
    skey, order = torch.sort(key, stable=True)
    last = torch.ones_like(skey, dtype=torch.bool)
    last[:-1] = skey[1:] != skey[:-1]
    return order[last]


def _prio(priority, device):
    return None if priority is None else priority.to(device=device, dtype=torch.int64).to(torch.int32).contiguous()   # uint32 bit pattern


def voxel_sample_point_major(pts_pm: torch.Tensor, target: int, rotations: torch.Tensor = None, seed: int = None, priority=None) -> torch.Tensor:
    """One cloud [n,3] on the GPU -> int64 [target] ascending, through pps_voxel_sample_f32 (all rounds in one launch).
    rotations: float32 [R,3,3,3] (draw_rotations); priority: optional integer tensor [n] with values < 2^32."""
    n = pts_pm.shape[0]
    rot = (draw_rotations() if rotations is None else rotations).to(torch.float32).reshape(-1, 27).contiguous().to(pts_pm.device, non_blocking=True)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    pr = _prio(priority, pts_pm.device)
    out = torch.empty((target,), dtype=torch.int64, device=pts_pm.device)
    L = _lib.lib()
    st = torch.cuda.current_stream(pts_pm.device).cuda_stream
    if n > L.pps_voxel_sample_max_points():
        # beyond the LDS tables of the one-launch kernel: the same procedure with its tables in a device workspace (any cloud size stays on the GPU)
        nbytes = L.pps_voxel_sample_large_ws_bytes(n)
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=pts_pm.device)
        _lib.check(L.pps_voxel_sample_large_f32(pts_pm.data_ptr(), n, int(target), ctypes.c_float(-1.0), rot.data_ptr(), rot.shape[0],
                                                ctypes.c_uint32(seed & 0xffffffff), pr.data_ptr() if pr is not None else None, out.data_ptr(), None,
                                                ws.data_ptr(), nbytes, st), 'pps_voxel_sample_large_f32')
        return out
    _lib.check(L.pps_voxel_sample_f32(pts_pm.data_ptr(), n, int(target), ctypes.c_float(-1.0), rot.data_ptr(), rot.shape[0],
                                      ctypes.c_uint32(seed & 0xffffffff), pr.data_ptr() if pr is not None else None, out.data_ptr(), None, st),
               'pps_voxel_sample_f32')
    return out


def voxel_sample_batch_point_major(pts_bpm: torch.Tensor, target: int) -> torch.Tensor:
    """A batch of equally sized clouds [b,n,3] on the GPU -> int64 [b,target], one launch (one workgroup per cloud)."""
    b, n = pts_bpm.shape[0], pts_bpm.shape[1]
    rot = draw_rotations(batch=b).reshape(b, -1, 27).contiguous().to(pts_bpm.device, non_blocking=True)
    seed = random.getrandbits(31)
    out = torch.empty((b, target), dtype=torch.int64, device=pts_bpm.device)
    _lib.check(_lib.lib().pps_voxel_sample_batch_f32(pts_bpm.data_ptr(), b, n, int(target), rot.data_ptr(), rot.shape[1],
                                                     ctypes.c_uint32(seed & 0xffffffff), None, out.data_ptr(), None,
                                                     torch.cuda.current_stream(pts_bpm.device).cuda_stream), 'pps_voxel_sample_batch_f32')
    return out


def sampling_quantized(pts_batch, ratio=None, n_support=None, support_points=None, support_points_ids=None, _rotations=None, _priority=None):
    """Voxel-stratified random sub-sampling to exactly max(1, int(N*ratio)) points (poco_data_loader.py:59-134):
    voxel edge = bbox diagonal / sqrt(n), three random axis rotations, one point per occupied voxel, remove, halve, repeat;
    the last round is truncated at random.  Stochastic (python `random`, torch RNG) like the reference.
    GPU clouds go through the HIP kernels (ids ascending; one launch with LDS tables up to pps_voxel_sample_max_points() points, the
    workspace variant beyond); CPU tensors (host-logic tests) through the torch-op loop below, which consumes `random` / the torch generator exactly like the reference and
    returns its ids in its order (tests/test_driver_parity_cpu.py against tests/golden/sampling.npz).
    Test hooks shared by both paths: `_rotations` [R,3,3,3]; `_priority` integer [N] (< 2^32): the last round keeps the
    representatives with the smallest priority instead of drawing a permutation."""
    if support_points is not None:
        return support_points, support_points_ids
    assert (ratio is None) != (n_support is None)
    b, _, n = pts_batch.shape
    target = max(1, int(n * ratio)) if ratio is not None else n_support
    if target == n:
        ids = torch.arange(n, dtype=torch.long, device=pts_batch.device).unsqueeze(0).expand(b, n)
        return pts_batch, ids
    if not 0 < target < n:
        raise ValueError('Search Quantized - ratio value error {} should be in ]0,1]'.format(ratio))
    if pts_batch.is_cuda and n >= 2:                 # every cloud size on the device (LDS kernel up to 10240 points, workspace kernel beyond)
        ids = torch.stack([voxel_sample_point_major(_point_major(pts_batch[i]), target, _rotations, priority=_priority) for i in range(b)], dim=0)
        return torch.gather(pts_batch, 2, ids.unsqueeze(1).expand(b, 3, target)), ids
    extent = pts_batch.max(dim=2)[0] - pts_batch.min(dim=2)[0]
    vox0 = extent.norm(2, dim=1) / math.sqrt(target)
    all_ids = []
    for i in range(b):
        pts = pts_batch[i].clone().transpose(0, 1)
        ids = torch.arange(pts.shape[0], device=pts.device)
        vox, count, picked = vox0[i], 0, []
        rnd = 0
        while True:
            mats = draw_rotations(1)[0] if _rotations is None else _rotations[rnd]
            rnd += 1
            rot = pts
            for a in range(3):
                rot = rot @ mats[a].t().to(pts.device, pts.dtype)
            perm = _representatives(rot, vox)
            if count + perm.shape[0] < target:
                picked.append(ids[perm])
                count += perm.shape[0]
                keep = torch.ones(ids.shape[0], dtype=torch.bool, device=pts.device)
                keep[perm] = False
                pts, ids = pts[keep], ids[keep]
                vox = vox / 2
            else:
                need = target - count
                if _priority is None:
                    sel = perm[torch.randperm(perm.shape[0])[:need].to(perm.device)]
                else:
                    cand = ids[perm]
                    pr = _priority.to(cand.device)[cand].to(torch.int64)
                    sel = perm[torch.argsort(pr * (n + 1) + cand)[:need]]
                picked.append(ids[sel])
                break
        all_ids.append(torch.cat(picked))
    ids = torch.stack(all_ids, dim=0)
    support = torch.gather(pts_batch, 2, ids.unsqueeze(1).expand(b, 3, ids.shape[1]))
    return support, ids


# Levels with at least this many points per cloud are searched with block culling.  Measured on 10 x 10 000-point clouds
# (round-2 probe time_id_tables.py, git history): arranging levels 0 and 1 costs 0.81 ms of small torch launches and the culling search of their five tables
# 1.00 ms against 1.36 ms for the exhaustive search -- with k = 16 the per-query merge networks, not the distance tests, bound both
# kernels, so culling only pays for larger clouds.  A 10k-point encoder pass therefore stays on the exhaustive kernel.
BLOCKED_MIN_POINTS = 16384


def _tables_batch(levels, segmentation=True, picks=None):
    """The 9 (+4) kNN tables of a batch of equally sized clouds from its 5 levels ([B,n_a,3] each) -> {name: int64 [B,m,k]}.
    picks[a] (int64 [B,n_{a+1}], or None): which points of level a became level a+1.  A support point IS a point of its level (same
    coordinates, bit for bit), so its neighbours within that level are the row of the level's own table: ids{a}{a+1} = ids{a}{a}[picks[a]]
    is a gather instead of a search (the reference searches again, poco_data_loader.py:171-186, and finds the same rows) -- 17 % of the
    distance tests of the 13 tables.
    Tables over large levels (>= BLOCKED_MIN_POINTS points per cloud) go through the block-culling
    search in ONE launch (pps_knn_blocked_batch_f32; their query sets are visited in Morton order), the small ones through the
    exhaustive pps_knn_multi_f32.  Same results either way (bit-identical indices, tests/test_gpu_sampling.py)."""
    nb = levels[0].shape[0]
    todo = []
    for a in range(5):
        todo.append(('ids{}{}'.format(a, a), a, a, 16))
        if a < 4:
            todo.append(('ids{}{}'.format(a, a + 1), a, a + 1, 16))
            if segmentation:
                todo.append(('ids{}{}'.format(a + 1, a), a + 1, a, 1))
    derived = []
    if picks is not None:
        derived = [(a, name) for name, pa, qa, k in todo for a in [pa] if qa == pa + 1 and k == 16 and picks[a] is not None]
        names = {name for _, name in derived}
        todo = [t for t in todo if t[0] not in names]
    blocked = {a: ops.BlockedLevel(levels[a]) for a in range(5) if levels[a].shape[1] >= BLOCKED_MIN_POINTS}
    big = [(name, pa, qa, k) for name, pa, qa, k in todo if pa in blocked]
    small = [(name, pa, qa, k) for name, pa, qa, k in todo if pa not in blocked]
    ret = {}
    if big:
        outs = ops.knn_blocked_batch([(blocked[pa], blocked.get(qa, levels[qa]), k) for _, pa, qa, k in big])
        for (name, _, _, _), o in zip(big, outs):
            ret[name] = o
    if small:
        ps, qs, ks = [], [], []
        for _, pa, qa, k in small:
            for b in range(nb):
                ps.append(levels[pa][b]); qs.append(levels[qa][b]); ks.append(min(k, levels[pa].shape[1]))
        outs = ops.knn_batch_point_major(ps, qs, ks)
        for i, (name, _, _, _) in enumerate(small):
            ret[name] = torch.stack(outs[i * nb:(i + 1) * nb], dim=0)
    for a, name in derived:
        own = ret['ids{}{}'.format(a, a)]
        ret[name] = torch.gather(own, 1, picks[a].unsqueeze(-1).expand(-1, -1, own.shape[2]))
    return ret


def get_fkaconv_ids(data, segmentation: bool = True):
    """4 support levels (ratio 0.25) and the 13 kNN tables of poco_data_loader.py:137-209."""
    pts = data['pts'].clone()
    unbatched = pts.dim() == 2
    if unbatched:
        pts = pts.unsqueeze(0)
    if pts.is_cuda and 4 <= pts.shape[2] <= _lib.lib().pps_voxel_sample_max_points():
        # fused GPU path, everything point-major: 4 sampling launches for the whole batch (one workgroup per cloud and level),
        # the 13 tables of up to 4 clouds per kNN launch
        nb = pts.shape[0]
        levels = [pts.transpose(1, 2).contiguous().float()]                  # [B,n,3]
        picks = []
        for _ in range(4):
            cur = levels[-1]
            n = cur.shape[1]
            target = max(1, int(n * 0.25))
            if target == n or n < 2:
                levels.append(cur)
                picks.append(None)
            else:
                ids = voxel_sample_batch_point_major(cur, target)
                levels.append(torch.gather(cur, 1, ids.unsqueeze(-1).expand(nb, target, 3)).contiguous())
                picks.append(ids)
        ret = {}
        for name, t in _tables_batch(levels, segmentation, picks).items():
            ret[name] = t.squeeze(0) if unbatched else t
        for a in range(1, 5):
            t = levels[a].transpose(1, 2).contiguous()
            ret['support{}'.format(a)] = t.squeeze(0) if unbatched else t
        ret['_levels_point_major'] = [[levels[a][b] for a in range(5)] for b in range(nb)]      # reused by FKAConvNetwork.forward_point_major
        return ret
    levels = [pts]
    for _ in range(4):
        levels.append(sampling_quantized(levels[-1], 0.25)[0])
    ret = {}
    squeeze = (lambda t: t.squeeze(0)) if unbatched else (lambda t: t)
    for a in range(5):
        ret['ids{}{}'.format(a, a)] = squeeze(knn(levels[a], levels[a], 16))
        if a < 4:
            ret['ids{}{}'.format(a, a + 1)] = squeeze(knn(levels[a], levels[a + 1], 16))
            if segmentation:
                ret['ids{}{}'.format(a + 1, a)] = squeeze(knn(levels[a + 1], levels[a], 1))
    for a in range(1, 5):
        ret['support{}'.format(a)] = squeeze(levels[a])
    return ret


def get_proj_ids(data, k: int):
    """poco_data_loader.py:212-240: accepts [B,3,N] or [B,N,3], batched or not."""
    pts, ptq = data['pts'], data['pts_query']
    unb_p, unb_q = pts.dim() == 2, ptq.dim() == 2
    if unb_p:
        pts = pts.unsqueeze(0)
    if unb_q:
        ptq = ptq.unsqueeze(0)
    if pts.shape[1] != 3:
        pts = pts.transpose(1, 2)
    if ptq.shape[1] != 3:
        ptq = ptq.transpose(1, 2)
    ids = knn(pts, ptq.to(pts.device), k, -1)
    if unb_p or unb_q:
        ids = ids.squeeze(0)
    return {'proj_ids': ids}


def get_data_poco(batch_data: dict):
    """poco_data_loader.py:243-270 (k=64 is hard-coded there, :261)."""
    fk = {'pts': torch.transpose(batch_data['pts_ms'], -1, -2), 'pts_query': torch.transpose(batch_data['pts_query_ms'], -1, -2)}
    if 'imp_surf_dist_ms' in batch_data:
        occ = torch.zeros_like(batch_data['imp_surf_dist_ms'], dtype=torch.int64)
        occ[torch.sign(batch_data['imp_surf_dist_ms']) > 0.0] = 1
        fk['occ'] = occ
    else:
        fk['occ'] = torch.zeros(fk['pts_query'].shape[:1])
    with torch.no_grad():
        net = get_fkaconv_ids(fk)
        net['proj_ids'] = get_proj_ids(fk, k=64)['proj_ids']
    batch_data.update(fk)
    batch_data.update(net)
    return batch_data


def normalize_patches(pts_local_ms, pts_query_ms):
    """ppsurf_data_loader.py:91-123 on device tensors: [Q,P,3], [Q,3] -> [Q,P,3]."""
    q, p = pts_local_ms.shape[0], pts_local_ms.shape[1]
    flat = pts_local_ms.reshape(q * p, 3).contiguous().float()
    idx = torch.arange(q * p, dtype=torch.int64, device=flat.device).view(q, p)
    return ops.patch_normalize(flat, pts_query_ms, idx, p)


def get_pts_local_ps(pts_raw_ms, pts_query, num_pts_local, idx=None):
    """poco_utils.py:67-72 on the GPU: P nearest raw points of each query, centred and scaled -> [Q,P,3].
    `idx` may carry an already computed neighbour table whose first P columns are the P nearest (same cloud)."""
    if idx is None:
        idx = (ops.knn_point_major(pts_raw_ms, pts_query, num_pts_local) if num_pts_local <= 64
               else ops.KnnBlocks(pts_raw_ms).query(pts_query, num_pts_local))
    return ops.patch_normalize(pts_raw_ms, pts_query, idx, num_pts_local)


def get_pts_local_ps_batch(raws, queries, num_pts_local, return_ms=False):
    """Patches of a fit batch: raws = list of B raw clouds [n_b,3] (sizes may differ), queries [B,Q,3] -> [B,Q,P,3].
    The B patch searches are one launch (pps_knn_multi_f32) when P <= 64.  return_ms: also the un-normalised neighbour
    coordinates `pts_local_ms` [B,Q,P,3] that the reference's dataset leaves in the batch (ppsurf_data_loader.py:83-89)."""
    b = len(raws)
    raws = [r.contiguous().float() for r in raws]
    qs = [queries[i].contiguous().float() for i in range(b)]
    if num_pts_local <= 64:
        ids = ops.knn_batch_point_major(raws, qs, [min(num_pts_local, r.shape[0]) for r in raws])
    else:
        ids = [ops.KnnBlocks(r).query(q, num_pts_local) for r, q in zip(raws, qs)]
    ps = torch.stack([ops.patch_normalize(raws[i], qs[i], ids[i], num_pts_local) for i in range(b)])
    if return_ms:
        return ps, torch.stack([raws[i][ids[i]] for i in range(b)])
    return ps
