"""N>1 path on CPU: 2 processes over gloo.  The sharded region growing must produce exactly the single-process volume
(the reference driver's golden volume), and the helper collectives must agree across ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import REPO, load_golden


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ppsurf_amd import sharding
    from ppsurf_amd.reconstruct import create_volume
    sharding.set_query_sharding(True)
    sharding.MIN_SHARD = 1                                   # tiny lists here: let every rank take its share
    g = load_golden('create_volume')
    seen = []

    def field(q):
        seen.append(q.shape[0])
        return (0.4 - torch.linalg.norm(q.double(), dim=1)).float()

    vol = create_volume(field, torch.from_numpy(g['pts_ids'].astype(np.int64)), int(g['resolution']), float(g['step']),
                        float(g['bmin_pad'])).numpy()
    t = sharding.max_over_ranks(1.0 + rank, 'cpu')
    lat = torch.full((5, 3), float(rank + 1))
    cnt = torch.ones(5)
    sharding.allreduce_latents(lat, cnt)
    ids = torch.arange(11, dtype=torch.float32)
    gathered = sharding.sharded_map(lambda x: x * 2, ids)
    np.savez(os.path.join(out_dir, 'r{}.npz'.format(rank)), vol=vol, t=t, lat=lat.numpy(), cnt=cnt.numpy(), gathered=gathered.numpy(),
             seen=np.array(seen).sum())
    dist.destroy_process_group()


def test_two_rank_sharded_region_growing(tmp_path):
    from ppsurf_amd.sharding import shard_range
    assert [shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert [shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    port = 29000 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ref = load_golden('create_volume')['volume']
    r0, r1 = np.load(tmp_path / 'r0.npz'), np.load(tmp_path / 'r1.npz')
    for r in (r0, r1):
        assert np.array_equal(np.isnan(r['vol']), np.isnan(ref))
        np.testing.assert_allclose(np.nan_to_num(r['vol'], nan=7.0), np.nan_to_num(ref, nan=7.0), rtol=0, atol=2e-7)
        assert r['t'] == 2.0 and (r['lat'] == 3.0).all() and (r['cnt'] == 2.0).all()
        assert np.array_equal(r['gathered'], np.arange(11) * 2.0)
    total = int((~np.isnan(ref[1:-1, 1:-1, 1:-1])).sum())
    assert abs(int(r0['seen']) - int(r1['seen'])) <= 64 and int(r0['seen']) + int(r1['seen']) <= int((~np.isnan(ref)).sum())


def _grad_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ppsurf_amd.sharding import GradBuckets
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 2))
    unused = torch.nn.Linear(3, 3)                                   # never part of the graph (POCO's cv5/bn5)
    params = list(net.parameters()) + list(unused.parameters())
    from ppsurf_amd.sharding import broadcast_buffers
    bn = torch.nn.BatchNorm1d(4)
    bn.running_mean.fill_(float(rank + 1)); bn.num_batches_tracked.fill_(rank + 5)
    broadcast_buffers(bn)
    assert float(bn.running_mean[0]) == 1.0 and int(bn.num_batches_tracked) == rank + 5      # float buffers from rank 0, ints untouched
    buckets = GradBuckets(params, n_buckets=3)
    x = torch.from_numpy(np.random.default_rng(5).standard_normal((8, 6)).astype(np.float32))
    y = torch.from_numpy(np.random.default_rng(6).integers(0, 2, 8))
    opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=0.1)
    out = {}
    for step in range(2):
        buckets.zero()
        lo, hi = rank * 4, rank * 4 + 4                              # each rank sees half of the batch
        torch.nn.functional.cross_entropy(net(x[lo:hi]), y[lo:hi]).backward()
        buckets.finish()
        out['g{}'.format(step)] = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy().copy()
        out['unused_none{}'.format(step)] = np.array([p.grad is None for p in unused.parameters()])
        opt.step()
    # a parameter that only SOME ranks touch: every rank must keep the same averaged gradient for it, none may skip it
    half = torch.nn.Linear(6, 2)
    with torch.no_grad():
        half.weight.fill_(0.5); half.bias.fill_(0.1)
    b2 = GradBuckets(list(net.parameters()) + list(half.parameters()), n_buckets=2)
    b2.zero()
    loss = torch.nn.functional.cross_entropy(net(x[:4]), y[:4])
    if rank == 0:
        loss = loss + half(x[:4]).pow(2).sum()
    loss.backward()
    b2.finish()
    # buckets summed over the ranks in bfloat16 (PPS_GRAD_BUCKET_DTYPE=bf16): half the bytes on the wire, fp32 buffers for the optimizer
    import copy
    net3 = copy.deepcopy(net)                                        # its own parameters: the hooks of the bucket sets above stay out of it
    b3 = GradBuckets(list(net3.parameters()), n_buckets=2, comm_dtype=torch.bfloat16)
    b3.zero()
    torch.nn.functional.cross_entropy(net3(x[lo:hi]), y[lo:hi]).backward()
    b3.finish()
    out['g_bf16'] = torch.cat([p.grad.reshape(-1) for p in net3.parameters()]).numpy().copy()
    net4 = copy.deepcopy(net)
    b4 = GradBuckets(list(net4.parameters()), n_buckets=2)
    b4.zero()
    torch.nn.functional.cross_entropy(net4(x[lo:hi]), y[lo:hi]).backward()
    b4.finish()
    out['g_f32'] = torch.cat([p.grad.reshape(-1) for p in net4.parameters()]).numpy().copy()
    # the split-graph fit (defer=True): the step body packs EVERY bucket itself (pack_all) so that the copies are part of the recorded graph, also when
    # bucket 0 holds parameters that never get a gradient (POCO's cv5 / bn5) and therefore never completes in the hooks; a replay then
    # only calls replayed() + finish()
    net5, unused5 = copy.deepcopy(net), torch.nn.Linear(3, 3)
    b5 = GradBuckets(list(net5.parameters()) + list(unused5.parameters()), n_buckets=3, defer=True)
    assert any(id(p) in {id(q) for q in unused5.parameters()} for p in b5.buckets[0])
    for step in range(3):
        b5.zero()
        torch.nn.functional.cross_entropy(net5(x[lo:hi]), y[lo:hi]).backward()
        b5.pack_all()
        assert all(b5.launched) and not b5.handles
        views = {v.data_ptr() for vs in b5.views for v in vs}
        assert all(p.grad is not None and p.grad.data_ptr() in views for p in net5.parameters())
        assert all(float(f.abs().sum()) > 0 for f in b5.flat[1:])
        if step == 2:
            b5.replayed(set(b5.touched))                                 # what fit does behind a replayed graph
        b5.finish()
        out['g_defer{}'.format(step)] = torch.cat([p.grad.reshape(-1) for p in net5.parameters()]).numpy().copy()
        out['defer_unused_none{}'.format(step)] = np.array([p.grad is None for p in unused5.parameters()])
    # all-reduce / backward overlap (fit.StagedStep): explicit buckets = the parameters each backward STAGE completes, the collective of bucket k issued
    # by the caller right behind stage k (reduce(k)), before the next stage starts; finish() only waits and averages
    net6 = copy.deepcopy(net)
    groups = [list(net6[4].parameters())[::-1], list(net6[2].parameters())[::-1], list(net6[0].parameters())[::-1]]
    b6 = GradBuckets(list(net6.parameters()), defer=True, groups=groups)
    b6.order_log = []
    assert [len(b) for b in b6.buckets] == [2, 2, 2]
    for step in range(2):
        b6.zero()
        h1 = net6[1](net6[0](x[lo:hi]))
        h1c = h1.detach().requires_grad_(True)
        h2 = net6[3](net6[2](h1c))
        h2c = h2.detach().requires_grad_(True)
        torch.nn.functional.cross_entropy(net6[4](h2c), y[lo:hi]).backward()
        assert all(p.grad is None for p in net6[2].parameters()) and all(p.grad is None for p in net6[0].parameters())
        b6.reduce(0)
        assert len(b6.handles) == 1 and b6.reduced == [True, False, False]
        h2.backward(h2c.grad)
        b6.reduce(1)
        h1.backward(h1c.grad)
        if step == 1:
            try:
                b6.reduced[1] = False
                b6.reduce(2)                                             # a bucket may not overtake its predecessor: every rank sends 0, 1, 2
                raise SystemExit('out-of-order reduce was accepted')
            except AssertionError:
                b6.reduced[1] = True
        b6.reduce(2)
        b6.reduce(2)                                                     # idempotent
        b6.finish()
        out['g_staged{}'.format(step)] = torch.cat([p.grad.reshape(-1) for p in net6.parameters()]).numpy().copy()
    assert b6.order_log == ['reduce0', 'reduce1', 'reduce2'] * 2
    try:
        GradBuckets(list(net6.parameters()), groups=groups[:2])
        raise SystemExit('groups that do not cover the parameters were accepted')
    except ValueError:
        pass
    out['half_none'] = np.array([p.grad is None for p in half.parameters()])
    out['half_g'] = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in half.parameters()]).numpy().copy()
    out['w'] = torch.cat([p.detach().reshape(-1) for p in params]).numpy()
    np.savez(os.path.join(out_dir, 'g{}.npz'.format(rank)), **out)
    dist.destroy_process_group()


def test_two_rank_gradient_buckets_equal_full_batch_gradients(tmp_path):
    """Shapes sharded over 2 ranks + bucketed all-reduce == the gradient of the mean loss over the whole batch; a parameter
    that takes no part in the graph keeps grad None (the optimizer must not touch it)."""
    port = 31000 + os.getpid() % 2000
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'g0.npz'), np.load(tmp_path / 'g1.npz')
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 2))
    unused = torch.nn.Linear(3, 3)
    w_unused0 = torch.cat([p.detach().reshape(-1) for p in unused.parameters()]).numpy().copy()
    x = torch.from_numpy(np.random.default_rng(5).standard_normal((8, 6)).astype(np.float32))
    y = torch.from_numpy(np.random.default_rng(6).integers(0, 2, 8))
    torch.nn.functional.cross_entropy(net(x), y).backward()
    full = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy()
    np.testing.assert_allclose(r0['g0'], full, rtol=1e-5, atol=1e-7)
    assert np.array_equal(r0['g0'], r1['g0']) and np.array_equal(r0['g1'], r1['g1']) and np.array_equal(r0['w'], r1['w'])
    assert r0['unused_none0'].all() and r0['unused_none1'].all()
    assert np.array_equal(r0['g_bf16'], r1['g_bf16'])                       # identical replicas ...
    # ... and the fp32 buckets of the same weights to bfloat16 precision (each rank's bucket is rounded once, the sum once more)
    assert r0['g_bf16'].dtype == np.float32 and np.abs(r0['g_bf16'] - r0['g_f32']).max() <= 2.0 ** -7 * np.abs(r0['g_f32']).max()
    assert not np.array_equal(r0['g_bf16'], r0['g_f32'])
    for step in range(3):                                                   # deferred collectives (split-graph fit), eager and "replayed"
        assert np.array_equal(r0['g_defer{}'.format(step)], r0['g_f32']) and np.array_equal(r1['g_defer{}'.format(step)], r0['g_f32'])
        assert r0['defer_unused_none{}'.format(step)].all() and r1['defer_unused_none{}'.format(step)].all()
    for step in range(2):                                                   # staged backward + reduce(k) between the stages == the plain bucketed step
        assert np.array_equal(r0['g_staged{}'.format(step)], r1['g_staged{}'.format(step)])
        np.testing.assert_allclose(r0['g_staged{}'.format(step)], r0['g_f32'], rtol=1e-6, atol=1e-8)
    # touched on rank 0 only: kept on BOTH ranks with the same averaged value (rank 0's gradient / 2)
    assert not r0['half_none'].any() and not r1['half_none'].any() and np.array_equal(r0['half_g'], r1['half_g']) and np.abs(r0['half_g']).max() > 0
    half = torch.nn.Linear(6, 2)
    with torch.no_grad():
        half.weight.fill_(0.5); half.bias.fill_(0.1)
    half(x[:4]).pow(2).sum().backward()
    np.testing.assert_allclose(r0['half_g'], torch.cat([p.grad.reshape(-1) for p in half.parameters()]).numpy() / 2, rtol=1e-6, atol=1e-7)
    assert np.array_equal(r0['w'][-w_unused0.size:], w_unused0)              # untouched by AdamW's weight decay


def _four_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ppsurf_amd import sharding
    sharding.set_query_sharding(True)
    out = {}
    for tag, n, floor in (('uneven', 1003, 1), ('floor', 1003, 400), ('one', 50, 400), ('empty', 0, 1)):
        seen = []

        def fn(x):
            seen.append(x.shape[0])
            return x[:, 0] * 3 + rank * 0              # value independent of the rank that computed it
        items = torch.arange(n, dtype=torch.float32).view(n, 1)
        res = sharding.sharded_map(fn, items, min_shard=floor) if n else sharding.sharded_map(fn, items.view(0, 1), min_shard=floor)
        out[tag] = res.numpy()
        out[tag + '_seen'] = np.array(sum(seen))
    lat = torch.full((7, 2), float(rank)); cnt = torch.full((7,), 1.0)
    sharding.allreduce_latents(lat, cnt)
    out['lat'], out['cnt'] = lat.numpy(), cnt.numpy()
    out['wm'] = np.array(sharding.weighted_mean_over_ranks(float(rank + 1) * (rank % 2), rank % 2, 'cpu'))      # ranks 0, 2 have no batches
    np.savez(os.path.join(out_dir, 'f{}.npz'.format(rank)), **out)
    dist.destroy_process_group()


def test_four_ranks_uneven_ranges_and_chunk_floor(tmp_path):
    """world_size 4 over gloo: contiguous ranges that differ by one item, the per-rank floor (a rank never decodes fewer than
    MIN_SHARD queries; the others only join the collective), empty lists, and the weighted validation mean."""
    from ppsurf_amd.sharding import shard_ranges
    assert shard_ranges(1003, 4, 1) == [(0, 251), (251, 502), (502, 753), (753, 1003)]
    assert shard_ranges(1003, 4, 400) == [(0, 502), (502, 1003), (1003, 1003), (1003, 1003)]
    assert shard_ranges(50, 4, 400) == [(0, 50), (50, 50), (50, 50), (50, 50)]
    assert shard_ranges(100_000, 8) [3] == (37500, 50000) and shard_ranges(20_000, 8)[2] == (20000, 20000)
    port = 33000 + os.getpid() % 2000
    mp.spawn(_four_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    r = [np.load(tmp_path / 'f{}.npz'.format(i)) for i in range(4)]
    for i in range(4):
        assert np.array_equal(r[i]['uneven'], np.arange(1003) * 3.0) and np.array_equal(r[i]['floor'], np.arange(1003) * 3.0)
        assert np.array_equal(r[i]['one'], np.arange(50) * 3.0) and r[i]['empty'].shape == (0,)
        assert (r[i]['lat'] == 6.0).all() and (r[i]['cnt'] == 4.0).all()
        assert float(r[i]['wm']) == (2.0 + 4.0) / 2
    assert [int(r[i]['uneven_seen']) for i in range(4)] == [251, 251, 251, 250]
    assert [int(r[i]['floor_seen']) for i in range(4)] == [502, 501, 0, 0]
    assert [int(r[i]['one_seen']) for i in range(4)] == [50, 0, 0, 0]


def test_dense_grid_blocks_shard_into_contiguous_z_slab_runs():
    """dense z-slabs of the (R+2)^3 grid as the sharding unit.  The dense query list of bench_workloads.dense_chunks is the grid in index
    order; sharding.shard_ranges cuts any list into contiguous ranges, so every rank owns a run of consecutive voxels (whole z-columns / x-slabs
    plus at most two partial ones), the ranges tile the block, and the coordinates follow poco_utils.py:212-213."""
    import bench_workloads as workloads
    from ppsurf_amd import sharding
    cloud = np.random.default_rng(0).uniform(-0.5, 0.5, (500, 3)).astype(np.float32)
    res, chunk = 31, 4000
    chunks, nblocks = workloads.dense_chunks(cloud, res, chunk, 'cpu', n_chunks=3)
    n = res + 2
    assert nblocks == n ** 3 // chunk and len(chunks) == 3 and all(c.shape == (chunk, 3) for c in chunks)
    step, bmin_pad, _ = workloads.grid_geometry(cloud, res)
    first = chunks[0]
    lin = np.arange(chunk)
    ijk = np.stack([lin // (n * n), (lin // n) % n, lin % n], axis=1).astype(np.float32)
    assert np.array_equal(first.numpy(), ijk * np.float32(step) + np.float32(bmin_pad))          # block 0 starts at voxel (0, 0, 0), z fastest
    for world in (2, 4, 8):
        ranges = sharding.shard_ranges(chunk, world, min_shard=256)
        assert ranges[0][0] == 0 and ranges[-1][1] == chunk and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        sizes = [hi - lo for lo, hi in ranges]
        assert max(sizes) - min(sizes) <= 1
        for lo, hi in ranges:                                        # consecutive voxels: the z index advances by one, wrapping at n
            z = np.round((first[lo:hi, 2].numpy() - np.float32(bmin_pad)) / np.float32(step)).astype(np.int64)
            assert np.all((np.diff(z) == 1) | (np.diff(z) == -(n - 1)))


def _single_worker(rank, world, port, out_dir):
    """ONE rank, PPS_SINGLE_RANK_COLLECTIVES=1: the multi-rank code paths with their collectives (gloo here, RCCL in tests/test_gpu_nccl_single_rank.py)."""
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', PPS_BACKEND='gloo')
    from ppsurf_amd import sharding
    out = {}
    os.environ['PPS_SINGLE_RANK_COLLECTIVES'] = '0'
    assert sharding.init_process_group() == (0, 1) and sharding.init_process_group() == (0, 1)      # idempotent
    out['multi_off'] = sharding.multi()
    os.environ['PPS_SINGLE_RANK_COLLECTIVES'] = '1'
    out['multi_on'] = sharding.multi()
    sharding.set_query_sharding(True)
    sharding.profile_collectives(True)
    seen = []
    g = sharding.sharded_map(lambda x: (seen.append(x.shape[0]), x * 3)[1], torch.arange(7, dtype=torch.float32))
    out['gathered'], out['seen'], out['calls'] = g.numpy(), seen, sharding.STATS['calls']
    lat, cnt = torch.full((4, 2), 2.0), torch.ones(4)
    sharding.allreduce_latents(lat, cnt)
    out['lat'] = lat.numpy()
    # gradient buckets with an EMPTY middle group (a frozen backward stage): three buckets stay three, the empty one sends nothing
    ps = [torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2, 2))]
    b = sharding.GradBuckets(ps, defer=True, groups=[[ps[0]], [], [ps[1]]])
    b.order_log = []
    b.zero()
    (ps[0].sum() * 2 + ps[1].sum() * 3).backward()
    for k in range(3):
        b.reduce(k)
    b.finish()
    out['n_buckets'], out['order'] = len(b.buckets), list(b.order_log)
    out['g0'], out['g1'] = ps[0].grad.numpy().copy(), ps[1].grad.numpy().copy()
    b.hold = True                                             # measurement switch: reduce(k) does nothing, finish() sends everything
    b.order_log = []
    b.zero()
    (ps[0].sum() + ps[1].sum()).backward()
    for k in range(3):
        b.reduce(k)
    out['held_order'] = list(b.order_log)
    b.finish()
    out['held_g0'] = ps[0].grad.numpy().copy()
    torch.save(out, os.path.join(out_dir, 'single.pt'))
    dist.destroy_process_group()


def test_one_rank_group_runs_the_multi_rank_paths_when_asked(tmp_path):
    port = 31000 + os.getpid() % 2000
    mp.spawn(_single_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    out = torch.load(tmp_path / 'single.pt', weights_only=False)
    assert out['multi_off'] is False and out['multi_on'] is True
    assert np.array_equal(out['gathered'], np.arange(7) * 3.0) and out['seen'] == [7] and out['calls'] == 1
    assert (out['lat'] == 2.0).all()
    assert out['n_buckets'] == 3 and out['order'] == ['reduce0', 'reduce1', 'reduce2']
    assert (out['g0'] == 2.0).all() and (out['g1'] == 3.0).all()
    assert out['held_order'] == [] and (out['held_g0'] == 1.0).all()


def test_parameter_stages_follow_the_encoder_module_not_a_name_substring():
    """stages were keyed on the substring 'encoder.' of parameter names.  A module whose encoder is found through the attribute gets
    its blocks sorted into stages; another module that merely has 'encoder.' inside a parameter name does not; empty stages stay in the list."""
    sys.path.insert(0, REPO)
    from ppsurf_amd import train_graph

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.cv0 = torch.nn.Linear(2, 2)
            self.resnetb20 = torch.nn.Linear(2, 2)
            self.fcout = torch.nn.Linear(2, 2)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = Enc()
            self.mlp = torch.nn.Linear(2, 2)

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.network = Net()

    m = Model()
    st = train_graph.parameter_stages(m)
    ids = lambda ps: {id(p) for p in ps}
    assert ids(st[2]) == ids(m.network.encoder.cv0.parameters()) and ids(st[1]) == ids(m.network.encoder.resnetb20.parameters())
    assert ids(st[0]) == ids(m.network.encoder.fcout.parameters()) | ids(m.network.mlp.parameters())
    assert train_graph.parameter_stages(m.network)[2] == st[2]                      # a network holding `.encoder` directly
    for p in m.network.encoder.resnetb20.parameters():
        p.requires_grad_(False)
    assert [len(g) for g in train_graph.parameter_stages(m)] == [4, 0, 2]           # the frozen stage stays, empty

    class Other(torch.nn.Module):                                                   # 'encoder.' in the name, no encoder module of ours
        def __init__(self):
            super().__init__()
            self.my_encoder = torch.nn.ModuleDict({'cv0': torch.nn.Linear(2, 2)})

    assert [len(g) for g in train_graph.parameter_stages(Other())] == [2, 0, 0]
