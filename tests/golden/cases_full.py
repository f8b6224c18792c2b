"""The fit batch of BASELINE config 3 at its full size (10 shapes x 10 000 points, 2000 queries per shape, 50-point patches) as a deterministic
function of seeds: shared by tests/golden/make_golden_train_full.py (which runs the REFERENCE on it and stores outputs + digests of the id tables)
and tests/test_gpu_configs.py (which rebuilds it instead of storing 30 MB of inputs).  Id tables and patches come from the oracle's exact kNN
(oracle/ppsurf_oracle.py, C restatement); the generator asserts that the reference's own kNN returns the same tables."""
import hashlib

import numpy as np
import torch

from ppsurf_amd.synthetic import make_cloud

B, N, Q, P = 10, 10000, 2000, 50


def digest(t):
    a = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def full_fit_batch():
    """-> (data dict in the reference's layout incl. proj_ids, occ int64 [B,Q]); CPU tensors."""
    from oracle import ppsurf_oracle as O
    rng = np.random.default_rng(20260930)
    clouds = [make_cloud(N, seed=300 + i) for i in range(B)]
    pts = torch.from_numpy(np.stack([c.T for c in clouds]).copy())
    sups, cur = [], pts
    for _ in range(4):
        m = max(1, int(cur.shape[2] * 0.25))
        sel = np.stack([np.sort(rng.choice(cur.shape[2], m, replace=False)) for _ in range(B)])
        cur = torch.stack([cur[b][:, torch.from_numpy(sel[b])] for b in range(B)]).contiguous()
        sups.append(cur)
    data = {'pts': pts}
    data.update(O.fkaconv_ids_from_supports(pts, sups))
    queries, patches, dist = [], [], []
    for i in range(B):
        qq = (clouds[i][rng.choice(N, Q, replace=False)] + rng.normal(0, 0.02, (Q, 3))).astype(np.float32)
        for _ in range(8):
            # a query whose 65 nearest points hold two EQUAL fp32 squared distances has no defined table (the reference's float64 kd-tree and the
            # (d2, index) order may disagree): such queries are moved a little until there is none (1-2 of 20 000 at these sizes)
            _, d2 = O.knn_point_major(clouds[i], qq, 65, return_d2=True)
            tied = np.nonzero((np.diff(d2, axis=1) <= 0).any(axis=1))[0]
            if tied.size == 0:
                break
            qq[tied] += rng.normal(0, 1e-3, (tied.size, 3)).astype(np.float32)
        else:
            raise RuntimeError('tie-free queries not found')
        queries.append(qq)
        patches.append(O.get_pts_local_ps(clouds[i], qq, P))
        dist.append((0.4 - np.linalg.norm(qq, axis=1)).astype(np.float32))
    data['pts_query'] = torch.from_numpy(np.stack(queries)).transpose(1, 2).contiguous()          # [B,3,Q] (get_data_poco)
    data['pts_local_ps'] = torch.from_numpy(np.stack(patches))
    data['proj_ids'] = O.knn(pts, data['pts_query'], 64)
    occ = (torch.sign(torch.from_numpy(np.stack(dist))) > 0).to(torch.int64)
    return data, occ


def table_digests(data):
    return {k: digest(v) for k, v in sorted(data.items()) if torch.is_tensor(v) and v.dtype == torch.int64}
