"""Datasets / data module of the predict and test paths, without trimesh / Lightning.

Mirrors the parts of source/occupancy_data_module.py:18-253, source/poco_data_loader.py:273-412 and
source/ppsurf_data_loader.py:11-141 that feed `predict_step` / `test_step`: dataset directory layout
(`04_pts_vis/<shape>.xyz.ply`, `05_query_pts|05_query_dist/<shape>.ply.npy`, `testset.txt`), single-file inputs with
bbox normalisation (source/base/math.py:111-126), the batch dictionary keys, batch size 1.
The patch search of the test path runs on the GPU (ppsurf_amd.spatial) in the main process instead of a CPU kd-tree in
DataLoader workers.  Training datasets (augmentation, sub-sampling, DDP sampler) are not built in this round.
"""
import os

import numpy as np
import torch

from . import meshio, spatial


def in_file_is_dataset(in_file: str):
    return os.path.splitext(in_file)[1].lower() == '.txt'


def get_set_files(in_file: str):
    if in_file_is_dataset(in_file):
        d = os.path.dirname(in_file)
        return os.path.join(d, 'trainset.txt'), os.path.join(d, 'valset.txt'), os.path.join(d, 'testset.txt')
    return in_file, in_file, in_file


def read_shape_list(shape_list_file: str):
    with open(shape_list_file) as f:
        return [x.strip() for x in f.readlines() if x.strip()]


def get_pc_file(in_file, shape_name):
    if in_file_is_dataset(in_file):
        return os.path.join(os.path.dirname(in_file), '04_pts_vis', shape_name + '.xyz.ply')
    return in_file


def load_shape_data_pc(in_file, padding_factor, shape_name, normalize=False):
    """occupancy_data_module.py:227-253 (without the debug PLY and the kd-tree)."""
    pts_file = get_pc_file(in_file, shape_name)
    pts = meshio.load_pts(pts_file)
    if pts.shape[1] > 3:
        nrm = pts[:, 3:6]
        normals = nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
        pts = pts[:, :3]
    else:
        normals = np.zeros_like(pts)
    if normalize:
        bb_min, bb_max = pts.min(axis=0), pts.max(axis=0)
        pts = (pts - (bb_min + bb_max) * 0.5) / (np.max(bb_max - bb_min) * (1.0 + padding_factor))
    return {'pts_ms': pts.astype(np.float32), 'normals_ms': normals.astype(np.float32), 'pc_file_in': pts_file}


class ReconstructionDataset(torch.utils.data.Dataset):
    """PPSurfReconstructionDataset (ppsurf_data_loader.py:126-141): the whole cloud, no sub-sampling."""

    def __init__(self, in_file, padding_factor, with_raw=True):
        self.in_file, self.padding_factor, self.with_raw = in_file, padding_factor, with_raw
        self.shape_names = read_shape_list(in_file) if in_file_is_dataset(in_file) else [in_file]

    def __len__(self):
        return len(self.shape_names)

    def _queries(self, name):
        d = os.path.dirname(self.in_file)
        fq = os.path.join(d, '05_query_pts', name + '.ply.npy')
        fd = os.path.join(d, '05_query_dist', name + '.ply.npy')
        if os.path.isfile(fq):
            return np.load(fq).astype(np.float32), np.load(fd).astype(np.float32)
        return np.empty((0, 3), dtype=np.float32), np.empty((0, 3), dtype=np.float32)

    def __getitem__(self, i):
        data = load_shape_data_pc(self.in_file, self.padding_factor, self.shape_names[i], normalize=not in_file_is_dataset(self.in_file))
        q, dist = self._queries(self.shape_names[i])
        item = {'pts_ms': torch.from_numpy(data['pts_ms']), 'normals_ms': torch.from_numpy(data['normals_ms']),
                'pc_file_in': data['pc_file_in'], 'pts_query_ms': torch.from_numpy(q), 'imp_surf_dist_ms': torch.from_numpy(dist),
                'shape_id': torch.tensor(i)}
        if self.with_raw:
            item['pts_raw_ms'] = item['pts_ms']
        return item


class TestDataset(ReconstructionDataset):
    """PPSurfDataset without augmentation (ppsurf_data_loader.py:61-81, test loader): full cloud sub-sampled to
    manifold_points, ground-truth queries, patches of the raw cloud, FKAConv id tables and occupancy labels."""

    def __init__(self, in_file, padding_factor, num_pts_local, manifold_points, seed, device):
        super().__init__(in_file, padding_factor, with_raw=False)
        self.num_pts_local, self.manifold_points, self.device = num_pts_local, manifold_points, device
        self.rng = np.random.RandomState(seed)

    def __getitem__(self, i):
        item = super().__getitem__(i)
        raw = item['pts_ms'].to(self.device)
        if self.manifold_points is not None:
            n = raw.shape[0]
            sel = self.rng.choice(np.arange(n), size=self.manifold_points, replace=n < self.manifold_points)
            item['pts_ms'] = item['pts_ms'][sel]
            item['normals_ms'] = item['normals_ms'][sel]
        q = item['pts_query_ms'].to(self.device)
        item['pts_local_ps'] = spatial.get_pts_local_ps(raw, q, self.num_pts_local)
        item = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in item.items()}
        return spatial.get_data_poco(item)


def _collate1(item):
    """default_collate for batch size 1."""
    return {k: (v if k.startswith('_') else v.unsqueeze(0) if torch.is_tensor(v) else [v]) for k, v in item.items()}


class PocoDataModule:
    def __init__(self, in_file, workers, use_ddp, padding_factor, seed, manifold_points, patches_per_shape, do_data_augmentation,
                 batch_size):
        self.in_file, self.workers, self.use_ddp, self.padding_factor, self.seed = in_file, workers, use_ddp, padding_factor, seed
        self.manifold_points, self.patches_per_shape = manifold_points, patches_per_shape
        self.do_data_augmentation, self.batch_size = do_data_augmentation, batch_size
        self.trainset, self.valset, self.testset = get_set_files(in_file)
        self.num_pts_local = None
        self.device = 'cuda'

    def predict_dataloader(self):
        ds = ReconstructionDataset(self.testset, self.padding_factor)
        return (_collate1(ds[i]) for i in range(len(ds)))

    def test_dataloader(self):
        ds = TestDataset(self.testset, self.padding_factor, self.num_pts_local, self.manifold_points, self.seed, self.device)
        return (_collate1(ds[i]) for i in range(len(ds)))

    def train_dataloader(self):
        raise NotImplementedError('training data pipeline (augmentation, DDP sampler) is not built in this round')

    val_dataloader = train_dataloader


class PPSurfDataModule(PocoDataModule):
    def __init__(self, num_pts_local, in_file, workers, use_ddp, padding_factor, seed, manifold_points, patches_per_shape,
                 do_data_augmentation, batch_size):
        super().__init__(in_file=in_file, workers=workers, use_ddp=use_ddp, padding_factor=padding_factor, seed=seed,
                         manifold_points=manifold_points, patches_per_shape=patches_per_shape,
                         do_data_augmentation=do_data_augmentation, batch_size=batch_size)
        self.num_pts_local = num_pts_local
