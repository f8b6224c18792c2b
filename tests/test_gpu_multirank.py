"""Two ranks on ONE GPU over gloo (rehearsal of the RCCL paths the driver runs on 8 GPUs): query/pass-sharded predict of
one shape must give the single-rank mesh inputs (same volume), shape-sharded predict must split the test set."""
import os
import subprocess
import sys

import numpy as np
import pytest

from golden_util import REPO

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, 'tests'))
import numpy as np, torch, torch.distributed as dist
from golden_util import filled_sd
from source.ppsurf_model import PPSurfModel
from ppsurf_amd import reconstruct, sharding
from ppsurf_amd.synthetic import make_cloud
rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
if world > 1:
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sharding.set_query_sharding(True)
torch.manual_seed(7)
model = PPSurfModel(pointnet_latent_size=256, output_names=['imp_surf_sign'], in_channels=3, out_channels=2, k=64, lambda_l1=0.0, debug=False,
                    in_file='c.npy', results_dir={out!r}, padding_factor=0.05, name='t', network_latent_size=256, gen_subsample_manifold_iter=2,
                    gen_subsample_manifold=1000, gen_resolution_global=17, num_pts_local=50, rec_batch_size=700, gen_refine_iter=0, workers=0)
model.network.load_state_dict(filled_sd('', key='ppsurf'))
model = model.to('cuda:0').eval()
model.shard_queries = world > 1
cloud = make_cloud(2500, seed=3)
pts_cf = torch.from_numpy(cloud.T.copy()).to('cuda:0')
lat = model.encode_latents(pts_cf)
assert torch.isfinite(lat).all()
# volume on FIXED latents (the latent loop is stochastic): both runs must agree exactly
fixed = torch.from_numpy(np.random.default_rng(1).standard_normal((2500, 256)).astype(np.float32)).to('cuda:0')
shape = {{'pts': pts_cf.unsqueeze(0), 'latents': fixed.t().unsqueeze(0)}}
field = reconstruct.OccupancyField(model.network, shape, torch.from_numpy(cloud).unsqueeze(0), 700, 50)
bmin, bmax = cloud.min(), cloud.max(); step = (bmax - bmin) / 16
ids = torch.from_numpy(((cloud - bmin) / step + 1).astype(np.int32).astype(np.int64)).to('cuda:0')
vol = reconstruct.create_volume(field, ids, 17, step, bmin - step).cpu().numpy()
np.save(os.path.join({out!r}, 'vol_w{{}}_r{{}}.npy'.format(world, rank)), vol)
np.save(os.path.join({out!r}, 'nq_w{{}}_r{{}}.npy'.format(world, rank)), np.array([field.n_queries, float(lat.abs().mean())]))
'''


def test_query_sharded_volume_equals_single_rank(tmp_path):
    script = tmp_path / 'run.py'
    script.write_text(SCRIPT.format(repo=REPO, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    subprocess.check_call([sys.executable, str(script)], env=env, timeout=600)
    subprocess.check_call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                           '--master-port', str(29600 + os.getpid() % 300), str(script)], env=env, timeout=600)
    v1 = np.load(tmp_path / 'vol_w1_r0.npy')
    a, b = np.load(tmp_path / 'vol_w2_r0.npy'), np.load(tmp_path / 'vol_w2_r1.npy')
    assert np.array_equal(np.isnan(a), np.isnan(v1)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b))
    assert np.array_equal(a, v1, equal_nan=True)                # chunk-invariant kernels: the sharded volume EQUALS the single-rank one
    n1 = np.load(tmp_path / 'nq_w1_r0.npy')[0]
    n2 = np.load(tmp_path / 'nq_w2_r0.npy')[0] + np.load(tmp_path / 'nq_w2_r1.npy')[0]
    assert n2 == n1                                                     # the two ranks decoded disjoint halves


@pytest.mark.parametrize('mode', ['default', 'bf16', 'poco'])
def test_two_rank_fit_keeps_the_replicas_identical(tmp_path, mode):
    """pps.py fit launched as 2 ranks (gloo on one GPU here, RCCL on the 8-GPU node): shapes sharded by the DistributedSampler
    rule, bucketed gradient all-reduce; after training both replicas hold bit-identical parameters.
    mode 'bf16' (PPS_GRAD_BUCKET_DTYPE=bf16): buckets summed over the ranks in bfloat16, replicas still identical.
    mode 'poco': the POCO model -- its encoder.cv5 / bn5 never get a gradient and sit in the first gradient bucket, which therefore never completes
    inside backward; the replayed step must still pack and average every bucket (before the fix the replicas diverged silently)."""
    import yaml
    import torch
    from ppsurf_amd.synthetic import write_dataset
    from test_gpu_cli import BASE, PPS, OPT
    in_file = write_dataset(str(tmp_path / 'ds'), n_shapes=4, n_pts=2000, n_query=200)
    cfg = dict(BASE); cfg.update(OPT)
    paths = []
    model_name = 'poco_mini' if mode == 'poco' else 'ppsurf_mini'
    for name, c in (('poco', cfg), ('pps', {} if mode == 'poco' else PPS), ('mini', {'model': {'init_args': {'name': model_name}},
                                                          'data': {'init_args': {'in_file': in_file, 'batch_size': 1, 'use_ddp': True,
                                                                                 'manifold_points': 1000}},
                                                          'trainer': {'max_epochs': 4, 'precision': 'bf16-mixed'}})):
        paths += ['-c', str(tmp_path / (name + '.yaml'))]
        yaml.safe_dump(c, open(paths[-1], 'w'))
    script = tmp_path / 'run.py'
    script.write_text("import os, sys, torch\nsys.path.insert(0, {r!r})\nfrom ppsurf_amd import runner\n"
                      "m = runner.main(['pps.py', 'fit'] + {a!r})\n"
                      "torch.save({{k: v.cpu() for k, v in m.state_dict().items()}}, os.path.join({o!r}, 'sd_r' + os.environ['RANK'] + '.pt'))\n"
                      .format(r=REPO, a=paths, o=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', PPS_BACKEND='gloo', PPS_FIT_ORDER_LOG='1')
    env.update({'bf16': {'PPS_GRAD_BUCKET_DTYPE': 'bf16'}}.get(mode, {}))

    def launch(e, port):
        return subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                               '--master-port', str(port), str(script)], env=e, cwd=str(tmp_path), timeout=900, capture_output=True, text=True)
    out = launch(env, 29900 + os.getpid() % 90)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-9000:]
    # several ranks: forward + backward replayed as a HIP graph after three eager steps, collectives and optimizer eager behind it
    assert 'HIP-graph replay of the step: 1 graph(s) captured' in out.stdout and 'FAILED' not in out.stdout, out.stdout[-1500:]
    a, b = torch.load(tmp_path / 'sd_r0.pt'), torch.load(tmp_path / 'sd_r1.pt')
    params = [k for k in a if 'running_' not in k and 'num_batches' not in k and 'norm_radius' not in k]
    assert all(torch.equal(a[k], b[k]) for k in params)                    # same parameters on both replicas
    assert any(not torch.equal(a[k], b[k]) for k in a if 'running_mean' in k)     # buffers are rank-local (different shapes)
    state = torch.load(tmp_path / 'models' / model_name / 'version_0' / 'checkpoints' / 'last.ckpt', map_location='cpu')
    assert state['global_step'] == 8                                       # 4 shapes / 2 ranks / batch 1 x 4 epochs
    # all-reduce / backward overlap (fit.StagedStep): the backward pass runs in three stages -- eagerly for the first steps, then as three replayed
    # sub-graphs -- and bucket k's collective is issued right behind stage k, i.e. BEFORE the next stage starts, on both ranks in the same order
    import json
    logs = [json.load(open(tmp_path / 'models' / model_name / 'version_0' / 'order_rank{}.json'.format(r))) for r in (0, 1)]
    assert logs[0] == logs[1]
    eager = ['stage0', 'reduce0', 'stage1', 'reduce1', 'stage2', 'reduce2']
    replay = ['replay0', 'reduce0', 'replay1', 'reduce1', 'replay2', 'reduce2']
    steps = [logs[0][i:i + 6] for i in range(0, len(logs[0]), 6)]
    assert len(steps) == 8 and all(st in (eager, replay) for st in steps), steps
    assert steps[0] == eager and steps[-1] == replay and sum(st == replay for st in steps) >= 4
