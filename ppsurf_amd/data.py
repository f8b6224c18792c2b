"""Datasets / data module (predict, test, fit) without trimesh / Lightning.

Mirrors the parts of source/occupancy_data_module.py:18-253, source/poco_data_loader.py:273-412 and
source/ppsurf_data_loader.py:11-141 that feed `predict_step` / `test_step`: dataset directory layout
(`04_pts_vis/<shape>.xyz.ply`, `05_query_pts|05_query_dist/<shape>.ply.npy`, `testset.txt`), single-file inputs with
bbox normalisation (source/base/math.py:111-126), the batch dictionary keys, batch size 1.
The patch search runs on the GPU (ppsurf_amd.spatial) in the main process instead of a CPU kd-tree in DataLoader workers.
Fit batches are assembled the same way: the dataset items are plain arrays (sub-sampled cloud, augmented queries), and one
`collate_on_device` call per batch does patches, support levels and the 13 id tables of every shape on the device -- the work
the reference spreads over `workers` CPU processes (configs/device_server.yaml uses 48 of them for 4 GPUs).
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
        normals = np.zeros(pts.shape, dtype=np.float64)            # the reference's zeros_like of trimesh's float64 vertices
    if normalize:
        bb_min, bb_max = pts.min(axis=0), pts.max(axis=0)
        pts = (pts - (bb_min + bb_max) * 0.5) / (np.max(bb_max - bb_min) * (1.0 + padding_factor))
    # pts are cast to float32 after normalisation, normals stay float64 (occupancy_data_module.py:236-241; tests/golden/batch_manifest.json)
    return {'pts_ms': pts.astype(np.float32), 'normals_ms': normals.astype(np.float64), 'pc_file_in': pts_file}


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
        if self.num_pts_local is not None:                      # PocoDataset has no patches (poco_data_loader.py:344-396)
            item['pts_local_ps'] = spatial.get_pts_local_ps(raw, q, self.num_pts_local)
        item = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in item.items()}
        return spatial.get_data_poco(item)


def random_rotation_matrix(rand3) -> np.ndarray:
    """Uniform random rotation from three uniform numbers (Shoemake's method; the algorithm behind
    trimesh.transformations.random_rotation_matrix(rand), used at poco_data_loader.py:333 / ppsurf_data_loader.py:66;
    trimesh itself is not in the image) -> 3x3 float64."""
    r1, r2 = np.sqrt(1.0 - rand3[0]), np.sqrt(rand3[0])
    t1, t2 = 2.0 * np.pi * rand3[1], 2.0 * np.pi * rand3[2]
    w, x, y, z = np.cos(t2) * r2, np.sin(t1) * r1, np.cos(t1) * r1, np.sin(t2) * r2
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class TrainDataset(ReconstructionDataset):
    """PocoDataset / PPSurfDataset for fit and validation (poco_data_loader.py:273-396, ppsurf_data_loader.py:48-81):
    cloud sub-sampled to `manifold_points` (with replacement when it is smaller), ground-truth queries + signed distances,
    `patches_per_shape` random queries only under DDP (:384-389), random rotation of cloud, normals and queries when
    augmenting.  Like the reference, patches are searched in the UNROTATED raw cloud with the rotated queries
    (ppsurf_data_loader.py:62-73).  Items are host arrays; `collate_on_device` finishes a batch on the GPU."""

    def __init__(self, in_file, padding_factor, seed, use_ddp, manifold_points, patches_per_shape, do_data_augmentation, num_pts_local):
        super().__init__(in_file, padding_factor, with_raw=False)
        self.manifold_points, self.patches_per_shape = manifold_points, patches_per_shape
        self.do_data_augmentation, self.num_pts_local = do_data_augmentation, num_pts_local
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1))
        self.ddp = bool(use_ddp) and torch.cuda.device_count() > 1
        if self.ddp:
            import torch.distributed as dist
            if not dist.is_available() or not dist.is_initialized():
                raise RuntimeError('Requires distributed package to be available')
            seed += dist.get_rank()                                            # per-rank stream (:297-298)
        self.rng = np.random.RandomState(seed)

    def __getitem__(self, i):
        data = load_shape_data_pc(self.in_file, self.padding_factor, self.shape_names[i], normalize=not in_file_is_dataset(self.in_file))
        raw = data['pts_ms']
        pts, normals = raw, data['normals_ms']
        if self.manifold_points is not None:
            sel = self.rng.choice(np.arange(raw.shape[0]), size=self.manifold_points, replace=raw.shape[0] < self.manifold_points)
            pts, normals = raw[sel], normals[sel]
        q, dist_ = self._queries(self.shape_names[i])
        if self.ddp and self.patches_per_shape is not None and self.patches_per_shape > 0:
            sel = self.rng.choice(np.arange(q.shape[0]), self.patches_per_shape)
            q, dist_ = q[sel], dist_[sel]
        if self.do_data_augmentation:
            rot = random_rotation_matrix(self.rng.rand(3))
            pts = (pts @ rot.T).astype(np.float32)
            normals = (normals @ rot.T).astype(np.float32)           # trafo.transform_points(...).astype(np.float32), poco_data_loader.py:322-323
            q = (q @ rot.T).astype(np.float32)
        return {'pts_ms': pts, 'normals_ms': normals, 'pts_query_ms': q, 'imp_surf_dist_ms': dist_, 'pts_raw_ms': raw,
                'pc_file_in': data['pc_file_in'], 'shape_id': i}

    def collate_on_device(self, items, device, extras=True):
        """default_collate + the per-shape work of the reference's __getitem__ (patches, get_data_poco), on the device.
        extras: also the flat ids + CSR of the id tables that the BACKWARD pass of a training step uses (14 sorts) -- validation batches go through
        the fused inference path and never read them."""
        batch = {k: torch.from_numpy(np.stack([it[k] for it in items])).to(device, non_blocking=True)
                 for k in ('pts_ms', 'normals_ms', 'pts_query_ms', 'imp_surf_dist_ms')}
        batch['shape_id'] = torch.tensor([it['shape_id'] for it in items], device=device)
        batch['pc_file_in'] = [it['pc_file_in'] for it in items]
        if self.num_pts_local is not None:
            raws = [torch.from_numpy(it['pts_raw_ms']).to(device, non_blocking=True) for it in items]
            batch['pts_local_ps'], batch['pts_local_ms'] = spatial.get_pts_local_ps_batch(raws, batch['pts_query_ms'], self.num_pts_local,
                                                                                          return_ms=True)
        batch = spatial.get_data_poco(batch)
        if extras:
            from . import train_graph
            with torch.no_grad():
                batch.update(train_graph.table_extras(batch))   # flat ids + CSR of the id tables: built with the batch, not inside backward
        return batch


class PocoDataset(TrainDataset):
    """source/poco_data_loader.py:273-341 by name: fit / validation items without patches."""

    def __init__(self, in_file, padding_factor, seed, use_ddp, manifold_points, patches_per_shape, do_data_augmentation=True):
        super().__init__(in_file, padding_factor, seed, use_ddp, manifold_points, patches_per_shape, do_data_augmentation, None)


class PPSurfDataset(TrainDataset):
    """source/ppsurf_data_loader.py:48-123 by name, including its static patch helpers (device tensors or numpy arrays)."""

    def __init__(self, in_file, num_pts_local, padding_factor, seed, use_ddp, manifold_points, patches_per_shape, do_data_augmentation=True):
        super().__init__(in_file, padding_factor, seed, use_ddp, manifold_points, patches_per_shape, do_data_augmentation, num_pts_local)

    @staticmethod
    def get_patch_radii(pts_patch, query_pts):
        """:100-110: largest distance of a patch point from its query -> [Q]."""
        t = torch.as_tensor(pts_patch) - torch.as_tensor(query_pts).unsqueeze(1)
        r = torch.sqrt((t * t).sum(-1)).max(dim=1)[0]
        return r if torch.is_tensor(pts_patch) else r.numpy()

    @staticmethod
    def model_space_to_patch_space(pts_to_convert_ms, pts_patch_center_ms, patch_radius_ms):
        """:112-123: (p - centre) / radius."""
        p, c, r = torch.as_tensor(pts_to_convert_ms), torch.as_tensor(pts_patch_center_ms), torch.as_tensor(patch_radius_ms)
        out = (p - c.unsqueeze(1)) / r.reshape(-1, 1, 1)
        return out if torch.is_tensor(pts_to_convert_ms) else out.numpy()

    @staticmethod
    def normalize_patches(pts_local_ms, pts_query_ms):
        """:91-97.  CUDA tensors go through pps_patch_normalize_f32, anything else through the two helpers above."""
        if torch.is_tensor(pts_local_ms) and pts_local_ms.is_cuda:
            return spatial.normalize_patches(pts_local_ms, pts_query_ms)
        radius = PPSurfDataset.get_patch_radii(pts_local_ms, pts_query_ms)
        return PPSurfDataset.model_space_to_patch_space(pts_local_ms, pts_query_ms, radius)


def _record_stream(obj, stream):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


_LOADER_STREAMS = {}


def _loader_stream(device):
    """ONE side stream per device and process for the batch assembly, made at first use and kept.  How much the loader's kernels take from the
    optimisation step depends on WHICH stream of torch's pool it got: measured on the config-3 step (profiles/NOTES_r6.md section 2,
    tools/dbg/fit_after_legs.py), a loader stream that is the 11th, 15th or 19th stream made in the process runs its kernels at an equal share of
    the chip (3.7 ms elapsed per batch) and the step takes 21.6 ms, any other of the first twenty leaves them in the step's shadow (5.9 ms elapsed)
    and the step takes 19.8 ms -- a period of four in the stream index (the runtime's stream -> hardware queue map), same clocks, same box.  A
    DevicePrefetch used to take a NEW pool stream per epoch (DeviceBatchLoader.__iter__), so a long fit walked through the pool and every fourth
    epoch landed on the slow mapping; now the first choice -- in a `pps.py fit` process the first stream made, the fast regime -- holds for the
    whole run."""
    key = (device.type, device.index)
    st = _LOADER_STREAMS.get(key)
    if st is None:
        st = torch.cuda.Stream(device, priority=0)
        _LOADER_STREAMS[key] = st
    return st


class DevicePrefetch:
    """Builds the device side of the NEXT batch (patch search, support sampling, the 13 + 1 id tables: ~4 ms of kernels of which the sampling
    runs on 10 workgroups) on a second HIP stream while the optimisation step of the current batch runs on the main stream -- what the
    reference's DataLoader worker processes do on the CPU.  `take(make_current, make_next)` returns the current batch (built now if nothing was
    prefetched) and starts the next one.  The tensors are allocated on the side stream: before they are handed to the consumer the main
    stream waits for the side stream's event and every tensor is recorded on the main stream (caching-allocator rule for cross-stream use)."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.index is None:               # resolved HERE, on the constructing (training) thread: a loader thread starts on device 0
            self.device = torch.device('cuda', torch.cuda.current_device())
        # the loader's kernels are ~6 ms of GPU time per fit step, several of them chip-wide (the exhaustive kNN): they must not take workgroup slots
        # from the step they hide behind.  This device offers two stream priorities (torch.cuda.Stream.priority_range() == (0, -1)): the loader
        # gets the lower one -- which is the default level -- and the optimisation step runs on a high-priority stream (fit.step_stream)
        self.side = _loader_stream(self.device)
        self.pending = None                      # (batch, event)

    def launch(self, make, after_main=True):
        """Runs make() on the side stream -> (batch, event).  May be called from a loader thread (the stream context is thread-local);
        after_main: the side stream first waits for what the main stream has queued (inputs made there)."""
        if after_main:
            self.side.wait_stream(torch.cuda.current_stream(self.device))
        # a new thread starts on device 0: the HIP launches of the C ABI go to the CURRENT device, so it is set for this scope (rank r > 0)
        with torch.cuda.device(self.device), torch.cuda.stream(self.side):
            batch = make()
            ev = torch.cuda.Event()
            ev.record(self.side)
        return batch, ev

    def take(self, make_current, make_next):
        main = torch.cuda.current_stream(self.device)
        batch, ev = self.pending if self.pending is not None else self.launch(make_current)
        main.wait_event(ev)
        _record_stream(batch, main)
        self.pending = self.launch(make_next) if make_next is not None else None
        return batch

    def hand_over(self, batch, ev):
        """A batch built by launch() in another thread becomes usable on the calling thread's current stream."""
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ev)
        _record_stream(batch, main)
        return batch


class DeviceBatchLoader:
    """DataLoader stand-in for fit / validation: shuffling like torch's RandomSampler (non-DDP) or DistributedSampler
    (seed 0 + epoch, padded to a multiple of the world size, rank-strided; occupancy_data_module.py:108-137), batches built by
    TrainDataset.collate_on_device.  `set_epoch` has DistributedSampler semantics."""

    def __init__(self, dataset, batch_size, shuffle, device, rank=0, world_size=1):
        self.dataset, self.batch_size, self.shuffle, self.device = dataset, int(batch_size), shuffle, device
        self.rank, self.world_size, self.epoch = rank, world_size, 0
        self.table_extras = True                # build the backward pass's CSR tables with the batch (PocoDataModule switches it off for validation)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _indices(self):
        n = len(self.dataset)
        if self.world_size > 1:
            if self.shuffle:
                g = torch.Generator()
                g.manual_seed(0 + self.epoch)
                idx = torch.randperm(n, generator=g).tolist()
            else:
                idx = list(range(n))
            total = -(-n // self.world_size) * self.world_size
            if total > n:                                       # DistributedSampler: repeat the list as often as the padding needs
                idx = (idx * -(-total // max(n, 1)))[:total]
            return idx[self.rank:total:self.world_size]
        return torch.randperm(n).tolist() if self.shuffle else list(range(n))

    def __len__(self):
        n = len(self.dataset) if self.world_size == 1 else -(-len(self.dataset) // self.world_size)
        return -(-n // self.batch_size)

    def __iter__(self):
        """Host side of batch i+1 (file reads, sub-sampling, augmentation -- one background thread, items in order, so the
        dataset's random stream is consumed exactly as in a sequential loop) overlaps the device work of batch i."""
        import concurrent.futures
        idx = self._indices()
        starts = list(range(0, len(idx), self.batch_size))
        load = lambda s: [self.dataset[i] for i in idx[s:s + self.batch_size]]
        import os
        dev = torch.device(self.device)
        prefetch = DevicePrefetch(dev) if (dev.type == 'cuda' and os.environ.get('PPS_PREP_STREAM', '1') != '0') else None
        if not starts:                                   # a rank without batches (fewer shapes than ranks)
            return
        thread_build = prefetch is not None and getattr(self, 'thread_collate', True)
        with concurrent.futures.ThreadPoolExecutor(max_workers=1) as pool:
            # the plain file read of the first batch is submitted only by the branches that consume it: the loader-thread branch below reads
            # inside its own build job (a second read of batch 0 would also advance the dataset's random stream twice)
            nxt = None if thread_build else pool.submit(load, starts[0])
            if prefetch is None:
                for k, s in enumerate(starts):
                    items = nxt.result()
                    nxt = pool.submit(load, starts[k + 1]) if k + 1 < len(starts) else None
                    yield self.dataset.collate_on_device(items, self.device, self.table_extras)
                return
            # the loader thread also issues the device side of its batch (patches, support levels, id tables, their CSR) -- on a second stream,
            # so that it runs beside the optimisation step of the previous batch and its ~10 ms of host work are off the training thread.  The
            # thread stays two batches ahead; uploads and searches depend on nothing the main stream produces (after_main=False).
            if not getattr(self, 'thread_collate', True):
                # eager step (PPS_FIT_GRAPH=0): the training thread needs the interpreter for its ~1400 launches per step, a second thread issuing
                # the batch assembly starves it (36 instead of 27 ms per step) -- the assembly is then issued by the training thread itself,
                # still on the side stream and one batch ahead; the loader thread only reads files
                futs = {0: nxt}
                if len(starts) > 1:
                    futs[1] = pool.submit(load, starts[1])
                for k, s in enumerate(starts):
                    if k + 2 < len(starts):
                        futs[k + 2] = pool.submit(load, starts[k + 2])
                    cur, nxt_f = futs.pop(k), futs.get(k + 1)
                    yield prefetch.take(lambda: self.dataset.collate_on_device(cur.result(), self.device, self.table_extras),
                                        (lambda f=nxt_f: self.dataset.collate_on_device(f.result(), self.device, self.table_extras)) if nxt_f is not None else None)
                return
            build = lambda s: prefetch.launch(lambda: self.dataset.collate_on_device(load(s), self.device, self.table_extras), after_main=False)
            futs = {k: pool.submit(build, starts[k]) for k in range(min(2, len(starts)))}
            for k, s in enumerate(starts):
                batch, ev = futs.pop(k).result()
                if k + 2 < len(starts):
                    futs[k + 2] = pool.submit(build, starts[k + 2])
                yield prefetch.hand_over(batch, ev)


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

    def _fit_loader(self, set_file, augment, shuffle):
        import torch.distributed as dist
        ddp = bool(self.use_ddp) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        ds = TrainDataset(set_file, self.padding_factor, self.seed, self.use_ddp, self.manifold_points, self.patches_per_shape, augment,
                          self.num_pts_local)
        loader = DeviceBatchLoader(ds, self.batch_size, shuffle, self.device, dist.get_rank() if ddp else 0, dist.get_world_size() if ddp else 1)
        loader.table_extras = bool(augment or shuffle)          # the training loader; validation batches need no backward-pass tables
        return loader

    def train_dataloader(self):
        return self._fit_loader(self.trainset, self.do_data_augmentation, True)

    def val_dataloader(self):
        return self._fit_loader(self.valset, False, False)


class PPSurfDataModule(PocoDataModule):
    def __init__(self, num_pts_local, in_file, workers, use_ddp, padding_factor, seed, manifold_points, patches_per_shape,
                 do_data_augmentation, batch_size):
        super().__init__(in_file=in_file, workers=workers, use_ddp=use_ddp, padding_factor=padding_factor, seed=seed,
                         manifold_points=manifold_points, patches_per_shape=patches_per_shape,
                         do_data_augmentation=do_data_augmentation, batch_size=batch_size)
        self.num_pts_local = num_pts_local
