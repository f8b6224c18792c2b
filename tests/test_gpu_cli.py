"""The reference's command line on the native path: stacked YAML configs with the reference's structure, dotted overrides,
predict / test / rec subcommands on a synthetic dataset in the reference's directory layout."""
import os

import numpy as np
import pytest
import torch
import yaml

from golden_util import filled_sd

pytestmark = pytest.mark.gpu

BASE = {  # structure of configs/poco.yaml + configs/ppsurf.yaml (values shortened for a quick run)
    'seed_everything': 42,
    'trainer': {'accelerator': 'gpu', 'devices': -1, 'precision': '16-mixed', 'logger': False},
    'data': {'class_path': 'source.poco_data_loader.PocoDataModule',
             'init_args': {'use_ddp': False, 'in_file': 'unset', 'padding_factor': 0.05, 'seed': 42, 'manifold_points': 1500,
                           'patches_per_shape': -1, 'do_data_augmentation': True, 'batch_size': 10, 'workers': 0}},
    'model': {'class_path': 'source.poco_model.PocoModel',
              'init_args': {'output_names': ['imp_surf_sign'], 'in_channels': 3, 'out_channels': 2, 'k': 64, 'network_latent_size': 32,
                            'gen_subsample_manifold_iter': 2, 'gen_subsample_manifold': 10000, 'gen_resolution_global': 257,
                            'rec_batch_size': 50000, 'gen_refine_iter': 2, 'workers': 0, 'lambda_l1': 0.0, 'results_dir': 'results',
                            'name': 'poco', 'debug': False}}}
PPS = {'model': {'class_path': 'source.ppsurf_model.PPSurfModel',
                 'init_args': {'network_latent_size': 256, 'num_pts_local': 50, 'pointnet_latent_size': 256, 'debug': False}},
       'data': {'class_path': 'source.ppsurf_data_loader.PPSurfDataModule'}}


def _configs(tmp_path, in_file):
    base = str(tmp_path / 'poco.yaml'); pps = str(tmp_path / 'ppsurf.yaml'); mini = str(tmp_path / 'mini.yaml')
    yaml.safe_dump(BASE, open(base, 'w'))
    yaml.safe_dump(PPS, open(pps, 'w'))
    yaml.safe_dump({'model': {'init_args': {'name': 'ppsurf_mini', 'gen_resolution_global': 129, 'rec_batch_size': 25000}},
                    'data': {'init_args': {'in_file': in_file, 'batch_size': 10}}}, open(mini, 'w'))
    return ['-c', base, '-c', pps, '-c', mini]


def test_predict_and_test_subcommands(tmp_path, capsys):
    from ppsurf_amd import runner
    from ppsurf_amd.synthetic import write_dataset
    in_file = write_dataset(str(tmp_path / 'ds'), n_shapes=2, n_pts=2500, n_query=400)
    ckpt = str(tmp_path / 'last.ckpt')
    sd = {'network.' + k: v for k, v in filled_sd('', key='ppsurf').items()}
    torch.save({'state_dict': sd}, ckpt)                                  # Lightning checkpoint layout
    args = ['pps.py', 'predict'] + _configs(tmp_path, in_file) + ['--ckpt_path', ckpt, '--model.init_args.gen_resolution_global', '17',
                                                                   '--model.init_args.results_dir', str(tmp_path / 'res'), '--trainer.devices', '1']
    model = runner.main(args)
    assert model.gen_resolution_global == 17 and model.rec_batch_size == 25000 and model.name == 'ppsurf_mini'
    assert model.in_file == in_file and model.num_pts_local == 50          # argument links (poco.py:16-20, pps.py:25)
    out = capsys.readouterr().out
    mesh_dir = tmp_path / 'res' / 'ppsurf_mini' / 'ds' / 'meshes'
    n_mesh = len(list(mesh_dir.glob('*.ply'))) if mesh_dir.exists() else 0
    assert n_mesh + out.count('No reconstruction for') == 2
    runner.main(['pps.py', 'test'] + _configs(tmp_path, in_file) + ['--ckpt_path', ckpt])
    out = capsys.readouterr().out
    assert out.count('loss') == 2 and 'nan' not in out.split('loss')[1][:12]


def test_rec_rewrite_fit_and_cpu_are_refused(tmp_path):
    from ppsurf_amd import runner
    cloud = tmp_path / 'c.npy'
    np.save(cloud, np.zeros((10, 3), np.float32))
    rec = runner.handle_rec_subcommand(['pps.py', 'rec', str(cloud), str(tmp_path / 'out'), '--model.init_args.rec_batch_size', '25000'])
    assert rec[1] == 'predict' and 'configs/ppsurf_50nn.yaml' in rec and rec[-2:] == ['--model.init_args.rec_batch_size', '25000']
    with pytest.raises(ValueError):
        runner.handle_rec_subcommand(['pps.py', 'rec', str(tmp_path / 'missing.ply'), 'out'])
    with pytest.raises(NotImplementedError):
        runner.main(['pps.py', 'fit'] + _configs(tmp_path, 'x.txt'))
    with pytest.raises(RuntimeError, match='no CPU path'):
        runner.main(['pps.py', 'predict'] + _configs(tmp_path, 'x.txt') + ['--trainer.accelerator', 'cpu'])


def test_poco_model_predict_through_the_cli(tmp_path, capsys):
    """configs/poco.yaml alone selects PocoModel / PocoDataModule (latent 32, no patches): predict runs natively too."""
    from ppsurf_amd import runner
    from ppsurf_amd.synthetic import write_dataset
    in_file = write_dataset(str(tmp_path / 'ds'), n_shapes=1, n_pts=2000, n_query=100)
    base = str(tmp_path / 'poco.yaml')
    yaml.safe_dump(BASE, open(base, 'w'))
    model = runner.main(['pps.py', 'predict', '-c', base, '--data.init_args.in_file', in_file, '--model.init_args.gen_resolution_global', '17',
                         '--model.init_args.results_dir', str(tmp_path / 'res'), '--model.init_args.rec_batch_size', '2000'])
    assert type(model).__name__ == 'PocoModel' and model.num_pts_local is None and model.network_latent_size == 32
    out = capsys.readouterr().out
    mesh_dir = tmp_path / 'res' / 'poco' / 'ds' / 'meshes'
    assert (len(list(mesh_dir.glob('*.ply'))) if mesh_dir.exists() else 0) + out.count('No reconstruction for') == 1
