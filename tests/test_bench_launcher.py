"""bench.py as the driver calls it: `python bench.py --gpus N` with no launcher around it must become N ranks by itself."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from golden_util import REPO


def _bench():
    sys.path.insert(0, REPO)
    import bench
    return bench


def test_launch_command_is_one_rank_per_gpu_on_loopback():
    bench = _bench()
    cmd = bench.launch_cmd(4, ['--gpus', '4', '--steps', '7', '--spawn'], 29999)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29999'
    i = cmd.index(os.path.join(REPO, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '4', '--steps', '7']            # the re-executed script gets the user's arguments, minus --spawn
    assert 0 < bench.free_port() < 65536


def test_self_spawn_starts_the_ranks_and_passes_their_exit_code_through(tmp_path, monkeypatch):
    """The launcher itself, on the CPU: the re-executed script is replaced by a stub that reports what torch.distributed.run gave it."""
    bench = _bench()
    stub = tmp_path / 'stub.py'
    stub.write_text("import os, sys\n"
                    "open(os.path.join({!r}, 'rank' + os.environ['RANK']), 'w').write(' '.join([os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'], "
                    "os.environ['MASTER_ADDR'], os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '')] + sys.argv[1:]))\n"
                    "sys.exit(3 if '--fail' in sys.argv and os.environ['RANK'] == '1' else 0)\n".format(str(tmp_path)))
    monkeypatch.setattr(bench, '__file__', str(stub))
    assert bench.self_spawn(2, ['--gpus', '2', '--quick']) == 0
    for r in (0, 1):
        assert (tmp_path / 'rank{}'.format(r)).read_text().split() == ['2', str(r), '127.0.0.1', '0', '--gpus', '2', '--quick']
    assert bench.self_spawn(2, ['--gpus', '2', '--fail']) != 0


def test_a_mismatched_launcher_is_refused_not_ignored():
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='1')
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '4', '--quick'], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and 'WORLD_SIZE' in p.stderr


@pytest.mark.gpu
def test_plain_gpus_2_prints_one_line_with_replicas_strong_and_fit():
    """`python bench.py --gpus 2` exactly as the driver would type it (plus the single-GPU rehearsal switches): self-spawn, two ranks on
    cuda:0 over gloo, ONE JSON line from rank 0 with n_gpus 2 carrying the three measurements of an N > 1 run: the replica
    figure (`value`, per-rank rates, shapes/hour), `strong` (ONE shape by both ranks, PPS_SHARD=queries) and `fit` (the staged data-parallel
    step with its all-reduce accounting)."""
    env = dict(os.environ, PPS_BENCH_DDP_BATCH='4', PPS_MIN_SHARD='4096')
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--same-gpu', '--steps', '2', '--warmup', '1',
                        '--shapes', '1'], capture_output=True, text=True, timeout=1500, cwd=REPO, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['world_size_seen'] == 2 and d['steps'] == 2 and d['scaling'] == 'weak'
    assert d['config']['parallelism'].startswith('shape-level replicas x2')          # not "query-block sharding": that is d['strong']
    assert len(d['per_rank_queries_per_s']['all']) == 2 and d['per_rank_queries_per_s']['min'] > 0
    assert d['value'] > 0 and d['repeats'] >= 1 and d['timed_s'] >= 1.0
    assert d['shapes_per_hour'] and d['shapes_per_hour'] > 0 and len(d['reconstruction']['per_rank_shapes_per_hour']['all']) == 2
    assert 'cpu_baseline' not in d                                                   # rank 0 at N = 1 only
    st, ft = d['strong'], d['fit']
    assert st['scaling'] == 'strong' and st['value'] > 0 and st['shapes_per_hour'] > 0 and 10 <= st['collectives_per_shape'] <= 40
    assert 0 < st['collective_share_rank0'] < 1 and len(st['per_rank_decoder_queries_per_shape']) == 2
    assert abs(sum(st['per_rank_decoder_queries_per_shape']) - st['decoder_queries_per_shape']) < 1
    assert min(st['per_rank_decoder_queries_per_shape']) > 0.3 * st['decoder_queries_per_shape']      # both ranks decoded about half
    assert ft['ranks'] == 2 and ft['batch_per_rank'] == 2 and ft['global_batch'] == 4 and ft['graphs_captured'] == 3 and not ft['capture_failed']
    assert ft['ms_per_step'] > 0 and ft['allreduce_ms'] > 0 and ft['ms_per_step_collectives_behind_backward'] > 0
    assert 0.0 <= ft['overlap_share'] <= 1.0 and sum(ft['gradient_bytes_per_step']) == 4 * 13_749_111
    assert d['fit_ms_per_step'] == ft['ms_per_step']


@pytest.mark.gpu
def test_self_spawned_single_rank_gives_the_same_kind_of_line():
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--spawn', '--quick', '--steps', '3', '--warmup', '1'],
                       capture_output=True, text=True, timeout=900, cwd=REPO)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
    assert d['n_gpus'] == 1 and d['roofline']['frac'] > 0.2 and d['value'] > 1e6


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', ['f16x3', 'f32'])
def test_strong_scaling_line_reports_the_dtype_it_ran_and_its_collectives(dtype):
    """`bench.py --gpus 2 --scaling strong` (two gloo ranks on cuda:0): ONE shape reconstructed by both ranks together.  The line must carry the
    decoder dtype that actually ran (it said 'f32' whatever ran), the collective count per shape, and a positive rate."""
    env = dict(os.environ, PPS_MIN_SHARD='4096')                                    # the floor is an environment switch now
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--same-gpu', '--scaling', 'strong',
                        '--steps', '1', '--warmup', '1', '--dtype', dtype], capture_output=True, text=True, timeout=1500, cwd=REPO, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['scaling'] == 'strong' and d['n_gpus'] == 2 and d['dtype'] == dtype and dtype in ('f32', 'f16x3')
    assert d['value'] > 0 and d['shapes_per_hour'] > 0
    assert 10 <= d['collectives_per_shape'] <= 40                                   # growth rounds + refinement rounds + the latent all-reduce
    assert 0 <= d['collective_share_rank0'] < 1


@pytest.mark.gpu
@pytest.mark.parametrize('leg', ['config2', 'config5'])
def test_extra_config_legs_report_what_baseline_md_asks_for(leg):
    """config 2 (R = 129: q/s, s/shape) and config 5 (N = 250k, P = 200, Q = 25 000: q/s and the k = 200 search / patch gather
    kernels' rates) are keys of the default line; `--only <leg>` prints the same object alone (what the counter passes profile)."""
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--only', leg], capture_output=True, text=True, timeout=900, cwd=REPO)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
    assert d['queries_per_s'] > 1e5 and d['decoder_dtype'] == 'f16x3'
    if leg == 'config2':
        assert 0 < d['s_per_shape'] < 10 and d['decoder_queries_per_shape'] > 5e5 and d['vertices'] > 10000
    else:
        assert d['ms_per_step'] > 0 and set(d['stage_ms']) == {'interp_pool', 'pointnet_stn_rows', 'pointnet_stn_fc', 'pointnet_feat_rows', 'decode_tail'}
        g = d['gather_kernels']
        assert g['knn_blocked_k200']['ms'] > 0 and g['patch_normalize_p200']['algorithmic_GB_s'] > 10
        assert d['band_queries_of_the_shape'] > 5_000_000


@pytest.mark.gpu
def test_fit_leg_runs_in_a_process_of_its_own_and_says_where_the_time_goes():
    """`python bench.py --only fit` = what the default line spawns for its fit leg (pps.py fit is a process of its own; at the end of the bench process
    the same step measured 19.7 .. 21.5 ms on one box depending on what ran before): step time, per-queue busy time, loader waits, GPU state."""
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--only', 'fit'], capture_output=True, text=True, timeout=900, cwd=REPO)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    f = d['fit']
    assert 5 < d['fit_ms_per_step'] < 60 and f['steps_timed'] == 120 and np.isfinite(f['loss'])
    assert f['queue_busy_ms']['step']['mean'] > 0.8 * d['fit_ms_per_step'] and f['queue_busy_ms']['loader']['mean'] > 0
    assert f['loader_wait_ms']['mean'] < 2.0 and 'before' in f['gpu_state']
    assert f.get('roofline') is not None or 'roofline_note' in f
