"""The whole pipeline with LEARNED weights: `pps.py fit` on synthetic bumpy spheres (SDF-sign labels), then `pps.py predict`
from the written checkpoint.  Parity tests pin the arithmetic; this pins the plumbing end to end: the optimisation reduces the
loss, the checkpoint round-trips into the fused inference path, and the reconstructed surface passes through the input cloud."""
import json
import os

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu


def test_fit_then_reconstruct(tmp_path, monkeypatch):
    from ppsurf_amd import runner, meshio
    from ppsurf_amd.synthetic import write_dataset
    from test_gpu_cli import BASE, PPS, OPT
    monkeypatch.chdir(tmp_path)
    in_file = write_dataset(str(tmp_path / 'ds'), n_shapes=24, n_pts=6000, n_query=1000)
    cfg = dict(BASE); cfg.update(OPT)
    res, epochs = 33, 24
    files = []
    for name, c in (('poco', cfg), ('pps', PPS), ('run', {'model': {'init_args': {'name': 'demo', 'gen_resolution_global': res, 'rec_batch_size': 20000,
                                                                                  'gen_refine_iter': 5, 'gen_subsample_manifold': 3000}},
                                                          'data': {'init_args': {'in_file': in_file, 'batch_size': 8, 'manifold_points': 3000}},
                                                          'trainer': {'max_epochs': epochs, 'precision': 'bf16-mixed'},
                                                          'lr_scheduler': {'init_args': {'milestones': [16, 21]}}})):
        files += ['-c', str(tmp_path / (name + '.yaml'))]
        yaml.safe_dump(c, open(files[-1], 'w'))
    runner.main(['pps.py', 'fit'] + files)
    recs = [json.loads(l) for l in open(tmp_path / 'models' / 'demo' / 'version_0' / 'metrics.jsonl')]
    steps = [r for r in recs if 'step' in r]
    vals = [r for r in recs if 'loss/val/00_all' in r]
    assert len(steps) == epochs * 3
    first, last = steps[0], steps[-6:]
    assert np.mean([s['loss/train/00_all'] for s in last]) < 0.8 * first['loss/train/00_all']
    assert np.mean([s['metrics/train/accuracy'] for s in last]) > 0.68
    assert vals[-1]['loss/val/00_all'] < 0.62 < vals[0]['loss/val/00_all'] + 0.1           # eval path sees the trained weights
    ckpt = tmp_path / 'models' / 'demo' / 'version_0' / 'checkpoints' / 'last.ckpt'
    with open(tmp_path / 'ds' / 'testset.txt', 'w') as f:
        f.write('synth_000\nsynth_001\n')
    runner.main(['pps.py', 'predict'] + files + ['--ckpt_path', str(ckpt), '--model.init_args.results_dir', str(tmp_path / 'res')])
    mesh_dir = tmp_path / 'res' / 'demo' / 'ds' / 'meshes'
    meshes = sorted(os.listdir(mesh_dir))
    assert meshes == ['synth_000.xyz.ply', 'synth_001.xyz.ply']
    for m in meshes:
        v = meshio.read_ply_vertices(str(mesh_dir / m))[:, :3]
        cloud = meshio.read_ply_vertices(str(tmp_path / 'ds' / '04_pts_vis' / m))[:, :3]
        d = np.sqrt(((cloud[::10, None, :] - v[None, :, :]) ** 2).sum(-1)).min(axis=1)
        assert v.shape[0] > 2000 and np.median(d) < 2.0 / (res - 1), (v.shape, np.median(d))


def test_fit_and_reconstruct_real_abc_shapes(tmp_path, monkeypatch):
    """The same on REAL data: four shapes of the reference's abc_minimal set (tests/golden/abc_mini4: clouds sub-sampled to
    <= 15000 points, the reference's own query points and signed distances): over-fit three CAD shapes for 40 epochs, then
    reconstruct one of them and the held-out validation shape."""
    import shutil
    from golden_util import GOLDEN
    from ppsurf_amd import runner, meshio
    from test_gpu_cli import BASE, PPS, OPT
    monkeypatch.chdir(tmp_path)
    shutil.copytree(os.path.join(GOLDEN, 'abc_mini4'), tmp_path / 'abc')
    in_file = str(tmp_path / 'abc' / 'testset.txt')
    cfg = dict(BASE); cfg.update(OPT)
    res, epochs = 49, 40
    files = []
    for name, c in (('poco', cfg), ('pps', PPS), ('run', {'model': {'init_args': {'name': 'abc', 'gen_resolution_global': res, 'rec_batch_size': 30000,
                                                                                  'gen_refine_iter': 5, 'gen_subsample_manifold': 5000}},
                                                          'data': {'init_args': {'in_file': in_file, 'batch_size': 3, 'manifold_points': 5000}},
                                                          'trainer': {'max_epochs': epochs, 'precision': 'bf16-mixed', 'check_val_every_n_epoch': 10},
                                                          'lr_scheduler': {'init_args': {'milestones': [28, 36]}}})):
        files += ['-c', str(tmp_path / (name + '.yaml'))]
        yaml.safe_dump(c, open(files[-1], 'w'))
    runner.main(['pps.py', 'fit'] + files)
    recs = [json.loads(l) for l in open(tmp_path / 'models' / 'abc' / 'version_0' / 'metrics.jsonl')]
    steps = [r for r in recs if 'step' in r]
    assert len(steps) == epochs
    # real SDF-sign labels (half of the queries hug the surface): 0.52 -> ~0.70 after 40 steps, ~0.77 after 200 (tools/abc_curve.py)
    assert np.mean([s['metrics/train/accuracy'] for s in steps[-5:]]) > 0.64 > steps[0]['metrics/train/accuracy']
    assert np.mean([s['loss/train/00_all'] for s in steps[-5:]]) < 0.58 < steps[0]['loss/train/00_all']
    ckpt = tmp_path / 'models' / 'abc' / 'version_0' / 'checkpoints' / 'last.ckpt'
    runner.main(['pps.py', 'predict'] + files + ['--ckpt_path', str(ckpt), '--model.init_args.results_dir', str(tmp_path / 'res')])
    mesh_dir = tmp_path / 'res' / 'abc' / 'abc' / 'meshes'
    names = [l.strip() for l in open(in_file) if l.strip()]
    trained = names[1]                                                    # testset.txt = [held-out shape, first training shape]
    assert os.path.isfile(mesh_dir / (trained + '.xyz.ply'))
    v = meshio.read_ply_vertices(str(mesh_dir / (trained + '.xyz.ply')))[:, :3]
    cloud = meshio.read_ply_vertices(str(tmp_path / 'abc' / '04_pts_vis' / (trained + '.xyz.ply')))[:, :3]
    d = np.sqrt(((cloud[::5, None, :] - v[None, :, :]) ** 2).sum(-1)).min(axis=1)
    assert v.shape[0] > 1000 and np.median(d) < 2.5 / (res - 1), (v.shape, np.median(d), 1.0 / (res - 1))
