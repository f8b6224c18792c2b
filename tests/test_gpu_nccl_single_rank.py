"""Every collective call site of the N > 1 paths executed once on RCCL (torch.distributed backend 'nccl') -- on a 1-GPU box.

RCCL refuses two ranks on one device, but it accepts a ONE-rank communicator.  PPS_SINGLE_RANK_COLLECTIVES=1 makes a one-rank process group run
the multi-rank code (sharding.multi()): the padded all-gather of sharding.sharded_map, the latent-sum all-reduce, the buffer broadcast, the three
gradient-bucket all-reduces issued between the replayed backward stages (fp32 and bf16 buckets), the int32 MAX of the parameter mask, the float64
reductions of the timing helpers and the barrier.  With one rank every collective is the identity, so the results must EQUAL the same run over gloo
(and, for predict, the unsharded run): what is tested is that RCCL takes these calls -- dtypes, stream order next to HIP-graph replays, device_id
initialisation -- not arithmetic.  The 2-rank arithmetic is tests/test_gpu_multirank.py (gloo, two ranks on one GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from golden_util import REPO

pytestmark = pytest.mark.gpu

PREDICT = r'''
import os, sys
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, 'tests'))
import numpy as np, torch, torch.distributed as dist
from golden_util import filled_sd
from source.ppsurf_model import PPSurfModel
from ppsurf_amd import reconstruct, sharding
from ppsurf_amd.synthetic import make_cloud
tag = sys.argv[1]
torch.cuda.set_device(0)
if tag != 'plain':
    sharding.init_process_group('cuda:0')             # PPS_BACKEND from the environment; device_id for nccl
    assert dist.get_backend() == os.environ['PPS_BACKEND'] and dist.get_world_size() == 1 and sharding.multi()
    sharding.set_query_sharding(True)
import random
torch.manual_seed(7); random.seed(7); np.random.seed(7)      # the support sampling of an encoder pass draws from python's `random`
model = PPSurfModel(pointnet_latent_size=256, output_names=['imp_surf_sign'], in_channels=3, out_channels=2, k=64, lambda_l1=0.0, debug=False,
                    in_file='c.npy', results_dir={out!r}, padding_factor=0.05, name='t', network_latent_size=256, gen_subsample_manifold_iter=2,
                    gen_subsample_manifold=1000, gen_resolution_global=17, num_pts_local=50, rec_batch_size=700, gen_refine_iter=0, workers=0)
model.network.load_state_dict(filled_sd('', key='ppsurf'))
model = model.to('cuda:0').eval()
model.shard_queries = tag != 'plain'
cloud = make_cloud(2500, seed=3)
pts_cf = torch.from_numpy(cloud.T.copy()).to('cuda:0')
sharding.profile_collectives(True)
random.seed(11)
lat = model.encode_latents(pts_cf)                    # sharded: subsets dealt to the (one) rank, partial sums all-reduced per wave
assert torch.isfinite(lat).all()
fixed = torch.from_numpy(np.random.default_rng(1).standard_normal((2500, 256)).astype(np.float32)).to('cuda:0')
shape = {{'pts': pts_cf.unsqueeze(0), 'latents': fixed.t().unsqueeze(0)}}
field = reconstruct.OccupancyField(model.network, shape, torch.from_numpy(cloud).unsqueeze(0), 700, 50)
bmin, bmax = cloud.min(), cloud.max(); step = (bmax - bmin) / 16
ids = torch.from_numpy(((cloud - bmin) / step + 1).astype(np.int32).astype(np.int64)).to('cuda:0')
vol = reconstruct.create_volume(field, ids, 17, step, bmin - step).cpu().numpy()
secs = sharding.collective_seconds()
np.save(os.path.join({out!r}, 'vol_' + tag + '.npy'), vol)
np.save(os.path.join({out!r}, 'lat_' + tag + '.npy'), lat.cpu().numpy())
np.save(os.path.join({out!r}, 'stat_' + tag + '.npy'), np.array([sharding.STATS['calls'], sharding.STATS['items'], secs, field.n_queries], dtype=np.float64))
if tag != 'plain':
    assert sharding.max_over_ranks(1.5, 'cuda:0' if tag == 'nccl' else 'cpu') == 1.5          # float64 MAX on the backend
    assert sharding.weighted_mean_over_ranks(3.0, 2, 'cuda:0' if tag == 'nccl' else 'cpu') == 1.5
    dist.barrier()
    dist.destroy_process_group()
'''


def test_query_sharded_predict_runs_its_collectives_on_rccl(tmp_path):
    """sharded_map's padded all-gather (one per growth round) and the latent all-reduce on a one-rank RCCL communicator: the volume equals the
    unsharded one voxel for voxel, the latents equal the gloo run's (the sharded latent loop draws its subsets from a shared generator, so it is
    compared with the same loop over gloo, not with the unsharded loop)."""
    script = tmp_path / 'run.py'
    script.write_text(PREDICT.format(repo=REPO, out=str(tmp_path)))
    for tag in ('plain', 'gloo', 'nccl'):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', PPS_BACKEND=tag, PPS_SINGLE_RANK_COLLECTIVES='0' if tag == 'plain' else '1')
        env.pop('MASTER_PORT', None); env.pop('RANK', None); env.pop('WORLD_SIZE', None)
        p = subprocess.run([sys.executable, str(script), tag], env=env, timeout=600, capture_output=True, text=True)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-4000:]
    v0, vg, vn = (np.load(tmp_path / 'vol_{}.npy'.format(t)) for t in ('plain', 'gloo', 'nccl'))
    assert np.array_equal(vn, v0, equal_nan=True) and np.array_equal(vg, v0, equal_nan=True)
    assert np.array_equal(np.load(tmp_path / 'lat_nccl.npy'), np.load(tmp_path / 'lat_gloo.npy'))
    s0, sn = np.load(tmp_path / 'stat_plain.npy'), np.load(tmp_path / 'stat_nccl.npy')
    assert s0[0] == 0 and sn[0] >= 2 and sn[1] == sn[3] == s0[3]          # every growth round went through the all-gather; same queries decoded
    assert sn[2] > 0                                                      # HIP events around the collectives measured something


@pytest.mark.parametrize('buckets', ['f32', 'bf16'])
def test_staged_fit_runs_its_collectives_on_rccl(tmp_path, buckets):
    """`pps.py fit` as ONE rank of an RCCL group running the multi-rank step (fit.StagedStep): parameter / buffer broadcast, buffer broadcast per
    step, bucket k all-reduced between backward stage k and k + 1 -- eagerly for three steps, then between the replays of the three stage graphs
    -- the int32 mask MAX, the validation-loss reduction.  The trained parameters equal the same fit over gloo to the bit."""
    import yaml
    import torch
    from ppsurf_amd.synthetic import write_dataset
    from test_gpu_cli import BASE, PPS, OPT
    in_file = write_dataset(str(tmp_path / 'ds'), n_shapes=4, n_pts=2000, n_query=200)
    cfg = dict(BASE); cfg.update(OPT)
    paths = []
    for name, c in (('poco', cfg), ('pps', PPS), ('mini', {'model': {'init_args': {'name': 'ppsurf_mini'}},
                                                           'data': {'init_args': {'in_file': in_file, 'batch_size': 1, 'use_ddp': True, 'manifold_points': 1000}},
                                                           'trainer': {'max_epochs': 2, 'precision': 'bf16-mixed'}})):
        paths += ['-c', str(tmp_path / (name + '.yaml'))]
        yaml.safe_dump(c, open(paths[-1], 'w'))
    sds, logs = {}, {}
    for backend in ('gloo', 'nccl'):
        work = tmp_path / backend
        work.mkdir()
        script = work / 'run.py'
        script.write_text("import os, sys, torch\nsys.path.insert(0, {r!r})\nfrom ppsurf_amd import runner\n"
                          "import torch.distributed as dist\n"
                          "m = runner.main(['pps.py', 'fit'] + {a!r})\n"
                          "assert dist.is_initialized() and dist.get_backend() == {b!r} and dist.get_world_size() == 1\n"
                          "torch.save({{k: v.cpu() for k, v in m.state_dict().items()}}, os.path.join({o!r}, 'sd.pt'))\n"
                          .format(r=REPO, a=paths, o=str(work), b=backend))
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', PPS_BACKEND=backend, PPS_SINGLE_RANK_COLLECTIVES='1', PPS_FIT_ORDER_LOG='1')
        env.pop('MASTER_PORT', None); env.pop('RANK', None); env.pop('WORLD_SIZE', None)
        if buckets == 'bf16':
            env['PPS_GRAD_BUCKET_DTYPE'] = 'bf16'
        out = subprocess.run([sys.executable, str(script)], env=env, cwd=str(work), timeout=900, capture_output=True, text=True)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-6000:]
        assert 'HIP-graph replay of the step: 1 graph(s) captured' in out.stdout and 'FAILED' not in out.stdout, out.stdout[-1500:]
        sds[backend] = torch.load(work / 'sd.pt')
        logs[backend] = json.load(open(work / 'models' / 'ppsurf_mini' / 'version_0' / 'order_rank0.json'))
    assert all(torch.equal(sds['gloo'][k], sds['nccl'][k]) for k in sds['gloo'])
    eager = ['stage0', 'reduce0', 'stage1', 'reduce1', 'stage2', 'reduce2']
    replay = ['replay0', 'reduce0', 'replay1', 'reduce1', 'replay2', 'reduce2']
    steps = [logs['nccl'][i:i + 6] for i in range(0, len(logs['nccl']), 6)]
    assert logs['nccl'] == logs['gloo'] and len(steps) == 8                # 4 shapes / batch 1 x 2 epochs
    assert steps[0] == eager and steps[-1] == replay and all(st in (eager, replay) for st in steps)


def test_bench_single_rank_line_carries_replicas_strong_and_fit_on_rccl():
    """`bench.py --gpus 1 --spawn --single-rank-collectives` = the driver's `bench.py --gpus N` code path with N = 1 on RCCL: self-spawn under
    torch.distributed.run, init_process_group('nccl', device_id=...), and ONE line with the replica figure, the `strong` block (query-block
    sharding of one shape) and the `fit` block (staged data-parallel step) -- the three things a SCALE run records per N."""
    env = dict(os.environ, PPS_BENCH_DDP_BATCH='4')                          # B = 50 // 1 = 50 shapes would not be a test
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--spawn', '--single-rank-collectives', '--steps', '2',
                        '--warmup', '1', '--shapes', '1'], capture_output=True, text=True, timeout=1500, cwd=REPO, env=env)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['scaling'] == 'weak' and d['value'] > 1e6 and 'replicas' in d['config']['parallelism']
    assert d['backend'].startswith('RCCL')
    st, ft = d['strong'], d['fit']
    assert st['scaling'] == 'strong' and st['value'] > 0 and st['shapes_per_hour'] > 0 and 10 <= st['collectives_per_shape'] <= 40
    assert 0 < st['collective_share_rank0'] < 1 and st['gathered_queries_per_shape'] == st['decoder_queries_per_shape']
    assert ft['ranks'] == 1 and ft['batch_per_rank'] == 4 and ft['graphs_captured'] == 3 and not ft['capture_failed']
    assert ft['ms_per_step'] > 0 and ft['allreduce_ms'] > 0 and ft['ms_per_step_collectives_behind_backward'] > 0
    assert 0.0 <= ft['overlap_share'] <= 1.0 and len(ft['gradient_bytes_per_step']) == 3
    assert d['fit_ms_per_step'] == ft['ms_per_step'] and np.isfinite(ft['loss'])
