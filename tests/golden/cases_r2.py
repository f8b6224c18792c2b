"""Input generators shared by tests/golden/make_golden_r2.py (which runs the reference on them) and the tests (which
regenerate them instead of storing them)."""
import numpy as np
import torch

from ppsurf_amd.synthetic import make_cloud


def clustered_cloud(seed, centres=200, per=50, sigma=0.004):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-0.5, 0.5, (centres, 1, 3)) + rng.normal(0, sigma, (centres, per, 3))).reshape(-1, 3).astype(np.float32)


SAMPLING_CASES = [('bumpy10k', lambda: make_cloud(10000, seed=21), 31), ('bumpy2500', lambda: make_cloud(2500, seed=22), 32),
                  ('clustered10k', lambda: clustered_cloud(3), 33), ('tight4k', lambda: clustered_cloud(5, 40, 100, 0.0005), 34),
                  ('n625', lambda: make_cloud(625, seed=23), 35), ('n156', lambda: make_cloud(156, seed=24), 36),
                  ('n39', lambda: make_cloud(39, seed=25), 37), ('n9', lambda: make_cloud(9, seed=26), 38)]


def bumpy_field(q: torch.Tensor) -> torch.Tensor:
    """Analytic occupancy logit difference of the refinement fixture: > 0 inside a bumpy sphere; float32 torch ops only."""
    r = torch.sqrt((q * q).sum(-1))
    return 14.0 * (0.33 + 0.05 * torch.sin(9.0 * q[..., 0]) * torch.cos(7.0 * q[..., 1]) + 0.03 * torch.sin(11.0 * q[..., 2]) - r)


def forward_case(g):
    """The batch dictionary of the `ppsurf_forward` fixture (cloud by seed, support selections and queries from the file, id
    tables and patches rebuilt by the oracle's exact kNN): CPU tensors in the reference's layout."""
    from oracle import ppsurf_oracle as O
    cloud = make_cloud(10000, seed=71)
    pts = torch.from_numpy(cloud.T.copy()).unsqueeze(0)
    sups, cur = [], pts
    for i in (1, 2, 3, 4):
        cur = cur[:, :, torch.from_numpy(g['sel{}'.format(i)].astype(np.int64))].contiguous()
        sups.append(cur)
    data = {'pts': pts}
    data.update(O.fkaconv_ids_from_supports(pts, sups))
    q = g['query']
    data['pts_query'] = torch.from_numpy(q.T.copy()).unsqueeze(0)
    data['pts_local_ps'] = torch.from_numpy(O.get_pts_local_ps(cloud, q, 50)).unsqueeze(0)
    return data
