"""Synthetic MEASUREMENT workloads on top of the product path (bench.py, tools/): nothing here is used by predict / fit.

  band_chunks            the query list of the FIRST region-growing round of a cloud (every voxel within +-2 of a voxel that holds
                         an input point, poco_utils.py:181-213), cut into rec_batch_size chunks
  SteeredField           OccupancyField whose RESULT is replaced by the analytic occupancy of the synthetic shape after the real
                         kernels have decoded the chunk: formula-filled weights describe no surface, so region growing would stop
                         after one round; with the steering the query counts are those of a real shape of that geometry
  reconstruct_steered    one whole reconstruction (latent loop, region growing, Marching Cubes, clean-up, 10 refinement rounds)
  FitStep                one optimisation step at BASELINE config 3 (B shapes x 10k points, 2000 queries, P = 50)
  FitStepDDP             the same step as ppsurf_amd.fit runs it with several ranks (staged backward, bucket all-reduces between the stage replays)
  HipEvents              hipEvent_t handles for timing kernels INSIDE a C-ABI call (pps_decode_fwd_events_f32)
"""
import contextlib
import ctypes
import io
import time

import numpy as np
import torch

from ppsurf_amd import reconstruct, spatial, synthetic


class HipEvents:
    """n hipEvent_t created through the HIP runtime (the library torch has already loaded)."""
    _hip = None

    def __init__(self, n):
        if HipEvents._hip is None:
            HipEvents._hip = ctypes.CDLL('libamdhip64.so')
        self.n = n
        self.arr = (ctypes.c_void_p * n)()
        for i in range(n):
            ev = ctypes.c_void_p()
            if HipEvents._hip.hipEventCreate(ctypes.byref(ev)) != 0:
                raise RuntimeError('hipEventCreate failed')
            self.arr[i] = ev

    def elapsed_ms(self, i, j):
        ms = ctypes.c_float()
        rc = HipEvents._hip.hipEventElapsedTime(ctypes.byref(ms), ctypes.c_void_p(self.arr[i]), ctypes.c_void_p(self.arr[j]))
        if rc != 0:
            raise RuntimeError('hipEventElapsedTime failed ({})'.format(rc))
        return ms.value

    def __del__(self):
        try:
            for i in range(self.n):
                HipEvents._hip.hipEventDestroy(ctypes.c_void_p(self.arr[i]))
        except Exception:
            pass


def grid_geometry(cloud: np.ndarray, resolution: int, padding: int = 1):
    """poco_utils.py:52-61: scalar bounds, step, padded origin, voxel ids of the input points."""
    bmin, bmax = cloud.min(), cloud.max()
    step = (bmax - bmin) / (resolution - 1)
    bmin_pad = bmin - padding * step
    pts_ids = ((cloud - bmin) / step + padding).astype(np.int32).astype(np.int64)
    return step, bmin_pad, pts_ids


def band_chunks(cloud: np.ndarray, resolution: int, chunk: int, device, full_only: bool = True):
    """The first growth round of create_volume for `cloud`: all voxels of the (R+2)^3 grid within +-2 of an occupied voxel, in
    index order, as float32 coordinates, cut into chunks of `chunk` queries (only full chunks when full_only)."""
    step, bmin_pad, pts_ids = grid_geometry(cloud, resolution)
    n = resolution + 2
    ids = torch.from_numpy(pts_ids).to(device)
    seeds = torch.zeros((n, n, n), dtype=torch.bool, device=device)
    seeds[ids[:, 0], ids[:, 1], ids[:, 2]] = True
    coords = torch.nonzero(reconstruct._dilate(seeds, 2))
    q = coords.to(torch.float32) * np.float32(step) + np.float32(bmin_pad)
    chunks = [q[s:s + chunk].contiguous() for s in range(0, q.shape[0], chunk)]
    if full_only and chunks and chunks[-1].shape[0] < chunk:
        chunks.pop()
    return chunks, int(q.shape[0])


def dense_chunks(cloud: np.ndarray, resolution: int, chunk: int, device, n_chunks: int):
    """Dense-block queries ("evaluated at every voxel of the Marching-Cubes grid"): the (R+2)^3 grid of poco_utils.py:52-58 in
    index order (x slowest, z fastest) is cut into consecutive blocks of `chunk` voxels -- z-slab runs of the volume -- and `n_chunks` of them,
    evenly spaced over the volume, are returned as float32 coordinates `idx * step + bmin_pad` (poco_utils.py:212-213).  Returns (chunks, number
    of blocks in the whole grid)."""
    step, bmin_pad, _ = grid_geometry(cloud, resolution)
    n = resolution + 2
    total = n ** 3
    nblocks = total // chunk
    pick = np.unique(np.linspace(0, nblocks - 1, num=min(n_chunks, nblocks)).astype(np.int64))
    out = []
    for b in pick:
        lin = torch.arange(int(b) * chunk, int(b) * chunk + chunk, device=device, dtype=torch.int64)
        ijk = torch.stack([lin // (n * n), (lin // n) % n, lin % n], dim=1)
        out.append((ijk.to(torch.float32) * np.float32(step) + np.float32(bmin_pad)).contiguous())
    return out, nblocks


def csrc_digest():
    """sha1 over the kernel sources + headers: ties a committed counter file (profiles/*_pmc.json) to the code it was measured on."""
    import hashlib
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha1()
    d = os.path.join(here, 'ppsurf_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.h', '.cpp')):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), 'rb').read())
    return h.hexdigest()[:16]


def fit_digest():
    """csrc_digest() extended by the Python side of the fit step: ties profiles/*_train_pmc.json to the program it was measured on."""
    import hashlib
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha1(csrc_digest().encode())
    for name in ('train_graph.py', 'train_ops.py', 'fit.py', 'optim.py', 'modules.py', 'data.py'):
        h.update(open(os.path.join(here, 'ppsurf_amd', name), 'rb').read())
    return h.hexdigest()[:16]


class SteeredField(reconstruct.OccupancyField):
    norm = None                                     # (centre, scale) of the synthetic cloud, set by the caller

    def __call__(self, q):
        super().__call__(q)                         # the real kNN + patches + decoder; result discarded
        return synthetic.bumpy_occupancy(q, self.norm)


def make_model(resolution=257, p=50, chunk=50000, device='cuda:0'):
    from ppsurf_amd.lightning_api import PPSurfModel
    with contextlib.redirect_stdout(io.StringIO()):
        model = PPSurfModel(pointnet_latent_size=256, output_names=['imp_surf_sign'], in_channels=3, out_channels=2, k=64, lambda_l1=0.0,
                            debug=False, in_file='x.npy', results_dir='/tmp/res', padding_factor=0.05, name='bench', network_latent_size=256,
                            gen_subsample_manifold_iter=10, gen_subsample_manifold=10000, gen_resolution_global=resolution, num_pts_local=p,
                            rec_batch_size=chunk, gen_refine_iter=10, workers=1)
    model.network.load_state_dict(synthetic.network_state_dict('ppsurf', num_pts_local=p))
    return model.to(device).eval()


def reconstruct_steered(model, n_points=100_000, seed=42, device='cuda:0', return_mesh=False):
    """One reconstruction with the product's own driver (encode_latents + export_mesh_and_refine_vertices_region_growing_v3),
    every query decoded by the real kernels, growth steered by the analytic shape.  Returns a dict of seconds / counts."""
    cloud, norm = synthetic.make_cloud(n_points, seed=seed, return_norm=True)          # Gaussian noise sigma = 0.005, like the chunk bench
    cloud_t = torch.from_numpy(cloud).to(device)
    pts_cf = cloud_t.t().contiguous()
    SteeredField.norm = norm
    fields = []

    class _Field(SteeredField):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            fields.append(self)

    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    lat = model.encode_latents(pts_cf)
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    shape = {'pts': pts_cf.unsqueeze(0), 'latents': lat.t().unsqueeze(0)}
    old = reconstruct.FIELD_CLASS
    reconstruct.FIELD_CLASS = _Field
    try:
        mesh = reconstruct.export_mesh_and_refine_vertices_region_growing_v3(
            network=model.network, latent=shape, pts_raw_ms=cloud_t.unsqueeze(0), resolution=model.gen_resolution_global, padding=1,
            mc_value=0, num_pts=model.rec_batch_size, num_pts_local=model.num_pts_local, input_points=cloud,
            refine_iter=model.gen_refine_iter, out_value=1)
    finally:
        reconstruct.FIELD_CLASS = old
    torch.cuda.synchronize(device)
    t2 = time.perf_counter()
    verts, faces = mesh if mesh is not None else (np.zeros((0, 3)), np.zeros((0, 3)))
    out = {'latent_s': t1 - t0, 'surface_s': t2 - t1, 'total_s': t2 - t0, 'decoder_queries': fields[0].n_queries if fields else 0,
           'vertices': int(verts.shape[0]), 'faces': int(faces.shape[0])}
    if return_mesh:
        out['mesh'] = (verts, faces)
    return out


class FitStep:
    """BASELINE config 3 on one GPU: B shapes x 10 000 points, 2000 queries per shape, P = 50; id tables and patches are built
    by a loader thread on a second stream (what the reference's dataset workers do on the CPU; overlap_prep=False: inline), then forward,
    loss, backward, AdamW -- the step body of ppsurf_amd.fit (fused AdamW; graph=True replays it as a HIP graph like fit does by default)."""

    def __init__(self, batch=10, n=10000, q=2000, p=50, precision='bf16-mixed', device='cuda:0', n_batches=2, graph=False, overlap_prep=True):
        from ppsurf_amd import modules, fit, sharding
        self.p, self.dev = p, torch.device(device)
        with contextlib.redirect_stdout(io.StringIO()):
            net = modules.PPSurfNetwork(in_channels=3, latent_size=256, out_channels=2, k=64, num_pts_local=p, pointnet_latent_size=256)
        net.load_state_dict(synthetic.network_state_dict('ppsurf', num_pts_local=p))
        self.net = net.to(self.dev).train()
        from ppsurf_amd import optim
        self.opt = optim.AdamW(self.net.parameters(), lr=1e-3, eps=1e-5, weight_decay=1e-2, capturable=graph)   # configs/poco.yaml:60-69, the class ppsurf_amd.fit puts in for torch.optim.AdamW
        self.buckets = sharding.GradBuckets([q for q in self.net.parameters() if q.requires_grad])      # as ppsurf_amd.fit (one rank: no collective)
        self.autocast = {'bf16-mixed': torch.bfloat16, '16-mixed': torch.float16}.get(precision)
        self.scaler = torch.amp.GradScaler('cuda') if precision == '16-mixed' else None
        self.batches = [self._raw_batch(batch, n, q, s) for s in range(n_batches)]
        self.i = 0
        self.loss = None
        self.trace = None                    # start_trace(): per-step host wait for the loader + HIP events around the step / the loader's kernels

        class _Log:
            values = {}

        self.stepper = fit.GraphedStep(self._body, _Log(), enabled=graph)
        from ppsurf_amd import data
        self.prefetch = data.DevicePrefetch(self.dev) if overlap_prep else None
        self.fut = None
        if self.prefetch is not None:
            import concurrent.futures
            self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)

    def _raw_batch(self, b, n, q, seed):
        rng = np.random.default_rng(seed)
        pts, qry, dist = [], [], []
        for i in range(b):
            c = synthetic.make_cloud(n, seed=seed * 100 + i)
            qq = (c[rng.choice(n, q)] + rng.normal(0, 0.02, (q, 3))).astype(np.float32)
            pts.append(c); qry.append(qq); dist.append((0.4 - np.linalg.norm(qq, axis=1)).astype(np.float32))
        return {'pts_ms': torch.from_numpy(np.stack(pts)).to(self.dev), 'pts_query_ms': torch.from_numpy(np.stack(qry)).to(self.dev),
                'imp_surf_dist_ms': torch.from_numpy(np.stack(dist)).to(self.dev)}

    def _body(self, batch, bi):
        from ppsurf_amd import train_graph
        self.buckets.zero()
        with torch.autocast('cuda', dtype=self.autocast or torch.bfloat16, enabled=self.autocast is not None):
            logits = self.net.forward(batch)
            loss = torch.nn.functional.cross_entropy(logits.float(), batch['occ'], reduction='none').mean()
        if self.scaler is not None:                     # 16-mixed: loss scaling as ppsurf_amd.fit does it (scale / overflow flag stay on the device)
            self.scaler.scale(loss).backward()
            self.buckets.finish()
            self.scaler.step(self.opt)
            self.scaler.update()
        else:
            loss.backward()
            self.buckets.finish()
            self.opt.step()
        train_graph.release_step_caches()
        self.stepper.metrics.values = {'loss': loss.detach()}

    def _prepare(self, i):
        """Patches, support levels and id tables of batch i on the device (what data.TrainDataset.collate_on_device does)."""
        batch = dict(self.batches[i % len(self.batches)])
        b = batch['pts_ms'].shape[0]
        batch['pts_local_ps'] = spatial.get_pts_local_ps_batch([batch['pts_ms'][j] for j in range(b)], batch['pts_query_ms'], self.p)
        batch = {k: v for k, v in spatial.get_data_poco(batch).items() if not k.startswith('_')}
        from ppsurf_amd import train_graph
        with torch.no_grad():
            batch.update(train_graph.table_extras(batch))       # like data.TrainDataset.collate_on_device
        return batch

    def _build(self, i):
        tr = self.trace
        if tr is None:
            return self.prefetch.launch(lambda: self._prepare(i), after_main=False)

        def make():                                   # the loader's kernels between two events on ITS stream
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            t0 = time.perf_counter()
            batch = self._prepare(i)
            tr['loader_host_ms'].append((time.perf_counter() - t0) * 1e3)
            b.record()
            tr['loader_ev'].append((a, b))
            return batch
        return self.prefetch.launch(make, after_main=False)

    def start_trace(self):
        """From now on every call records: how long the host waited for the loader thread (`loader_wait_ms`), the step's own queue between the
        hand-over of the batch and the end of the step (`step_queue_ms`: static-input copy + graph replay, by HIP events on the step's stream), how
        long the step's stream waited for the loader's last kernel (`batch_wait_ms`: event before the hand-over to event after it), and the
        loader's kernels on its side stream (`loader_queue_ms`) with the host time of issuing them (`loader_host_ms`)."""
        self.trace = {'loader_wait_ms': [], 'step_ev': [], 'loader_ev': [], 'loader_host_ms': [], 'call_ms': []}

    def read_trace(self):
        tr, self.trace = self.trace, None
        torch.cuda.synchronize()
        if self.fut is not None:                      # the batch built ahead by the traced _build: its events are complete now
            self.fut.result()
            torch.cuda.synchronize()
        med = lambda v: float(np.median(v)) if len(v) else None
        mean = lambda v: float(np.mean(v)) if len(v) else None
        wait = [a.elapsed_time(b) for a, b, _ in tr['step_ev']]
        busy = [b.elapsed_time(c) for _, b, c in tr['step_ev']]
        loader = [a.elapsed_time(b) for a, b in tr['loader_ev']]
        return {'steps': len(busy), 'call_ms': {'mean': mean(tr['call_ms']), 'median': med(tr['call_ms']), 'max': max(tr['call_ms'])},
                'queue_busy_ms': {'step': {'mean': mean(busy), 'median': med(busy), 'max': max(busy)},
                                  'loader': {'mean': mean(loader), 'median': med(loader), 'max': max(loader) if loader else None}},
                'batch_wait_ms': {'mean': mean(wait), 'median': med(wait), 'max': max(wait)},
                'loader_wait_ms': {'mean': mean(tr['loader_wait_ms']), 'median': med(tr['loader_wait_ms']), 'max': max(tr['loader_wait_ms'])},
                'loader_host_ms': {'mean': mean(tr['loader_host_ms']), 'median': med(tr['loader_host_ms'])},
                'note': 'per step: call_ms = host time of one FitStep call; queue_busy_ms.step = HIP events on the step\'s stream from the hand-over of '
                        'the batch to the end of the replayed graph (the step\'s own kernels, back to back); batch_wait_ms = the step\'s stream waiting for '
                        'the loader\'s last kernel; loader_wait_ms = the HOST blocked on the loader thread; queue_busy_ms.loader = the loader\'s kernels '
                        'on the side stream (they share the GPU with the step), loader_host_ms = host time the loader thread needs to issue them'}

    def __call__(self):
        tr = self.trace
        t_call = time.perf_counter()
        if self.prefetch is None:
            batch = self._prepare(self.i)
        else:                                         # like data.DeviceBatchLoader: a loader thread builds the next batch on the side stream
            if self.fut is None:
                self.fut = self.pool.submit(self._build, self.i)
            t0 = time.perf_counter()
            batch, ev = self.fut.result()
            if tr is not None:
                tr['loader_wait_ms'].append((time.perf_counter() - t0) * 1e3)
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            self.fut = self.pool.submit(self._build, self.i + 1)
            batch = self.prefetch.hand_over(batch, ev)
        self.i += 1
        if tr is not None and self.prefetch is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
        self._step(batch)
        if tr is not None and self.prefetch is not None:
            e2 = torch.cuda.Event(enable_timing=True)
            e2.record()
            tr['step_ev'].append((e0, e1, e2))
            tr['call_ms'].append((time.perf_counter() - t_call) * 1e3)
        return self._loss()

    def _step(self, batch):
        self.stepper.run(batch, self.i)

    def _loss(self):
        return self.stepper.metrics.values['loss']

    def close(self):
        if self.prefetch is not None:
            if self.fut is not None:
                self.fut.result()
            self.pool.shutdown(wait=True)


class FitStepDDP(FitStep):
    """BASELINE config 3 as `pps.py fit` runs it under torch.distributed.run (one rank per GPU): the Lightning module's training_step, the backward
    pass in train_graph.N_STAGES stages -- eager for the first steps, then one replayed HIP graph per stage -- with gradient bucket k all-reduced
    (RCCL) between stage k and stage k + 1 (fit.StagedStep + sharding.GradBuckets(defer=True, groups=parameter_stages)), the buffer broadcast of
    DDP before the step, the per-parameter mask collective and the fused AdamW behind it.  Every rank builds its own batches (rank-offset seeds)."""
    WARMUP_STEPS = 8                                  # 3 eager + capture + first replays

    def __init__(self, batch=10, n=10000, q=2000, p=50, precision='bf16-mixed', device='cuda:0', n_batches=2, rank=0):
        from ppsurf_amd import fit, sharding, train_graph, optim, data
        self.p, self.dev = p, torch.device(device)
        self.model = make_model(257, p, 50000, device).train()
        params = [t for t in self.model.parameters() if t.requires_grad]
        self.opt = optim.AdamW(params, lr=1e-3, eps=1e-5, weight_decay=1e-2, fused=True, capturable=False)   # as fit() builds it for several ranks
        if sharding.multi():
            import torch.distributed as dist
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                dist.broadcast(t.data, src=0)
        self.buckets = sharding.GradBuckets(params, defer=True, groups=train_graph.parameter_stages(self.model))
        ctx, use_scaler = fit.autocast_context(precision, 'cuda')
        self.scaler = torch.amp.GradScaler('cuda', enabled=use_scaler)
        self.metrics = fit._MetricLog()
        self.model.__dict__['_fit_log'] = self.metrics
        self.core = fit.StagedStep(self.model, self.buckets, self.scaler, ctx, self.metrics, enabled=True)
        self.batches = [self._raw_batch(batch, n, q, 1000 * (rank + 1) + s) for s in range(n_batches)]
        self.i, self.trace, self.fut = 0, None, None
        self.prefetch = data.DevicePrefetch(self.dev)
        import concurrent.futures
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)

    def _step(self, batch):
        from ppsurf_amd import sharding, train_graph
        import os
        if os.environ.get('PPS_BENCH_DDP_TRACE') != '1':
            sharding.broadcast_buffers(self.model)    # DDP's broadcast_buffers=True: rank 0's BatchNorm statistics / norm_radius before the step
            self.core.run(batch, self.i)
            self.buckets.finish()                     # waits for the bucket all-reduces, averages, mask collective
            self.scaler.step(self.opt)
            self.scaler.update()
            train_graph.release_step_caches()
            return
        # diagnostic: host time of every phase with a device synchronisation behind it (PPS_BENCH_DDP_TRACE=1; not a timing mode)
        t = [time.perf_counter()]

        def mark():
            torch.cuda.synchronize()
            t.append(time.perf_counter())
        sharding.broadcast_buffers(self.model); mark()
        self.core.run(batch, self.i); mark()
        self.buckets.finish(); mark()
        self.scaler.step(self.opt); self.scaler.update(); mark()
        train_graph.release_step_caches()
        if sharding.world()[0] == 0:
            print('ddp step {}: broadcast_buffers {:.1f} ms, staged forward/backward + reduce issue {:.1f} ms, finish {:.1f} ms, optimizer {:.1f} ms'.format(
                self.i, *[(b - a) * 1e3 for a, b in zip(t[:-1], t[1:])]), flush=True)

    def _loss(self):
        return self.metrics.values['loss/train/00_all']

    def n_stage_graphs(self):
        return sum(len(e[1]) for e in self.core.graphs.values())

    def bucket_bytes(self):
        el = 4 if self.buckets.comm_dtype is None else torch.empty((), dtype=self.buckets.comm_dtype).element_size()
        return [int(f.numel()) * el for f in self.buckets.flat]

    def allreduce_alone_ms(self, reps=10, dist=None):
        """The three bucket all-reduces of a step issued back to back with nothing else on the GPU (what sharding.GradBuckets._all_reduce issues)."""
        if dist is None:
            return 0.0
        if self.fut is not None:                      # the loader's kernels of the batch built ahead are not part of this
            self.fut.result()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            hs = [self.buckets._all_reduce(bi) for bi in range(len(self.buckets.flat))]
            for h in hs:
                h.wait()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def close(self):
        super().close()
        self.model.__dict__.pop('_fit_log', None)
