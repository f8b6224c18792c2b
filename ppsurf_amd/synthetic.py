"""Deterministic synthetic inputs: formula-based parameter fill and point clouds.

No checkpoint of the reference travels with this repo (models/download_ppsurf_50nn.py needs
network).  Parity fixtures, tests and bench.py therefore regenerate every parameter from its
state-dict NAME and SHAPE alone, so both the reference (when imported in the build container
by tests/golden/make_golden.py) and this package fill byte-identical weights.

The cloud generator follows SURVEY.md 8(d): noisy sphere r=0.45 with low-frequency bumps,
normalised like source/base/math.py:111-126 (bbox centre -> 0, longest side * 1.05 -> 1).
"""
import zlib

import numpy as np


# layers whose output is not followed by a ReLU: unit gain so logits stay O(1)
_LINEAR_OUT = ('fc_query.weight', 'fc_value.weight', 'fc8.weight', 'point_net.conv3.weight', 'fcout.weight',
               'mlp.layers.2.0.weight', 'MLP.layers.2.0.weight')


_FKA_GAIN = 0.0015


def _rng(name: str, salt: int = 0) -> np.random.Generator:
    return np.random.default_rng(zlib.crc32(name.encode()) + 7919 * salt)


def fill_param(name: str, shape, salt: int = 0) -> np.ndarray:
    """float32 array for the state-dict entry `name` of `shape` (int64 scalar for counters)."""
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit('.', 1)[-1]
    rng = _rng(name, salt)
    if leaf == 'num_batches_tracked':
        return np.zeros(shape, dtype=np.int64)
    if leaf == 'running_mean':
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == 'running_var':
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == 'norm_radius':
        return rng.uniform(0.05, 0.15, shape).astype(np.float32)
    if leaf in ('alpha', 'beta'):
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == 'bias':
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if leaf == 'weight':
        if len(shape) == 1:  # norm affine scale
            return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        fan_in = int(np.prod(shape[1:]))
        gain = 2.0                                   # He: layers followed by a ReLU/SiLU keep O(1) activations
        if name.endswith('stn2.fc3.weight'):
            gain = 0.005                             # feature transform stays near identity (as trained STNs do)
        elif name.endswith('.cv.weight') and len(shape) == 4:
            gain = _FKA_GAIN                         # FKAConv kernel: the 16 neighbour weights add up, keep the U-Net O(1)
        elif any(name.endswith(e) for e in _LINEAR_OUT):
            gain = 1.0
        return (rng.standard_normal(shape) * np.sqrt(gain / fan_in)).astype(np.float32)
    raise KeyError('no fill rule for state-dict entry {!r}'.format(name))


def fill_state_dict(manifest, salt: int = 0) -> dict:
    """manifest: iterable of (name, shape) -> {name: np.ndarray}."""
    return {name: fill_param(name, shape, salt) for name, shape in manifest}


def network_state_dict(kind: str = 'ppsurf', salt: int = 0, num_pts_local: int = 50, as_torch: bool = True, quiet: bool = True) -> dict:
    """Formula-filled state dict of a whole network ('ppsurf': PPSurfNetwork(3,256,2,64,P,256), ppsurf_model.py:21-24 with
    configs/ppsurf.yaml; 'poco': PocoNetwork(3,32,2,64)).  Names and shapes come from this package's own parameter holders
    (identical to the reference's 455 / 363 entries, tests/test_host_api_cpu.py), so bench.py and smoke() need no fixture file."""
    import contextlib
    import io
    import torch
    from . import modules
    with contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext():
        net = (modules.PPSurfNetwork(3, 256, 2, 64, num_pts_local, 256) if kind == 'ppsurf' else modules.PocoNetwork(3, 32, 2, 64))
    sd = fill_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], salt)
    return {k: torch.from_numpy(v) for k, v in sd.items()} if as_torch else sd


def state_dict_digest(sd: dict) -> str:
    """Order-independent digest of a {name: array} dict (names, shapes, bytes)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(sd.keys()):
        a = np.ascontiguousarray(np.asarray(sd[name]))
        h.update(name.encode())
        h.update(str(a.shape).encode())
        h.update(str(a.dtype).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def normalize_cloud(pts: np.ndarray, padding_factor: float = 0.05) -> np.ndarray:
    """bbox centre -> origin, longest side * (1 + padding) -> 1 (source/base/math.py:111-126)."""
    bb_min = pts.min(axis=0)
    bb_max = pts.max(axis=0)
    centre = (bb_min + bb_max) * 0.5
    scale = float((bb_max - bb_min).max()) * (1.0 + padding_factor)
    return ((pts - centre) / scale).astype(np.float32)


def _bumpy_radius(d):
    """Radius of the synthetic shape along unit directions d [..,3] (numpy array or torch tensor)."""
    if isinstance(d, np.ndarray):
        sin, cos = np.sin, np.cos
    else:
        import torch
        sin, cos = torch.sin, torch.cos
    return 0.45 * (1.0 + 0.08 * sin(3.0 * d[..., 0]) * cos(2.0 * d[..., 1]) + 0.05 * sin(5.0 * d[..., 2]) + 0.04 * cos(4.0 * d[..., 0] + 1.0))


def make_cloud(n: int, seed: int = 42, noise: float = 0.005, return_norm: bool = False):
    """Bumpy sphere, float32 [n,3] inside the unit box.  return_norm: also (centre [3], scale) of the
    normalisation, for `bumpy_occupancy`."""
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = d * _bumpy_radius(d)[:, None] + noise * rng.standard_normal((n, 3))
    out = normalize_cloud(pts)
    if return_norm:
        bb_min, bb_max = pts.min(axis=0), pts.max(axis=0)
        return out, ((bb_min + bb_max) * 0.5, float((bb_max - bb_min).max()) * 1.05)
    return out


def bumpy_occupancy(q, norm):
    """Analytic stand-in occupancy of the make_cloud shape at normalised query points q [m,3] (torch tensor): > 0 inside,
    magnitude ~ distance to the surface along the ray from the centre.  Used to STEER region growing / refinement in timing
    workloads (formula-filled weights describe no surface); never a network output."""
    import torch
    centre, scale = norm
    p = q.double() * scale + torch.as_tensor(centre, dtype=torch.float64, device=q.device)
    r = torch.linalg.norm(p, dim=1).clamp_min(1e-12)
    return ((_bumpy_radius(p / r[:, None]) - r) / scale).to(torch.float32)


def make_band_queries(pts: np.ndarray, m: int, resolution: int = 257, seed: int = 1, spread: int = 2) -> np.ndarray:
    """m grid-node queries within +-`spread` voxels of cloud points of the (R+2)^3 MC grid.

    Mirrors the geometry of source/poco_utils.py:52-61,212-213: scalar bmin/bmax, step=(max-min)/(R-1),
    padding 1, coordinates idx*step + bmin_pad.  Sorted by voxel index (z fastest) like np.argwhere.
    """
    rng = np.random.default_rng(seed)
    bmin = float(pts.min())
    bmax = float(pts.max())
    step = (bmax - bmin) / (resolution - 1)
    bmin_pad = bmin - step
    sel = rng.integers(0, pts.shape[0], size=m)
    ids = ((pts[sel] - bmin) / step + 1).astype(np.int32)
    ids = ids + rng.integers(-spread, spread + 1, size=ids.shape)
    ids = np.clip(ids, 0, resolution + 1)
    key = (ids[:, 0].astype(np.int64) * (resolution + 2) + ids[:, 1]) * (resolution + 2) + ids[:, 2]
    ids = ids[np.argsort(key, kind='stable')]
    return (ids.astype(np.float32) * np.float32(step) + np.float32(bmin_pad)).astype(np.float32)


def make_latents(c: int, n: int, seed: int = 77) -> np.ndarray:
    """Seeded N(0,1) latent table, channel-first [1,c,n] float32 (decoder cost is value independent)."""
    return np.random.default_rng(seed).standard_normal((1, c, n)).astype(np.float32)


def write_dataset(root: str, n_shapes: int = 3, n_pts: int = 3000, n_query: int = 2000, seed: int = 0):
    """Synthetic dataset in the reference's directory layout (source/occupancy_data_module.py:34-71): `04_pts_vis/<name>.xyz.ply`,
    `05_query_pts|05_query_dist/<name>.ply.npy` ((n_query,3) / (n_query,) float32, signed distance > 0 outside), and
    train/val/test set lists.  Returns the path of testset.txt."""
    import os
    from . import meshio
    rng = np.random.default_rng(seed)
    names = ['synth_{:03d}'.format(i) for i in range(n_shapes)]
    for i, name in enumerate(names):
        cloud = make_cloud(n_pts, seed=seed + i)
        meshio.write_ply_points(os.path.join(root, '04_pts_vis', name + '.xyz.ply'), cloud)
        q = (cloud[rng.integers(0, n_pts, n_query)] + rng.normal(0, 0.02, (n_query, 3))).astype(np.float32)
        near = cloud[np.argmin(((q[:, None, :] - cloud[None, ::8, :]) ** 2).sum(-1), axis=1) * 8]
        dist = (np.linalg.norm(q, axis=1) - np.linalg.norm(near, axis=1)).astype(np.float32)     # radial proxy of the signed distance
        for sub, arr in (('05_query_pts', q), ('05_query_dist', dist)):
            os.makedirs(os.path.join(root, sub), exist_ok=True)
            np.save(os.path.join(root, sub, name + '.ply.npy'), arr)
    for fname in ('trainset.txt', 'valset.txt', 'testset.txt'):
        with open(os.path.join(root, fname), 'w') as f:
            f.write('\n'.join(names) + '\n')
    return os.path.join(root, 'testset.txt')
