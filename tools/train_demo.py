"""End-to-end sanity of the whole pipeline on synthetic shapes: pps.py fit on bumpy spheres (SDF-sign labels), then pps.py
predict with the trained checkpoint -> meshes.  Shows the training path learns (accuracy, loss) and that a reconstruction with
learned weights closes a surface.    python tools/train_demo.py [--shapes 40] [--epochs 30] [--res 65]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import yaml

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', type=int, default=40)
    ap.add_argument('--epochs', type=int, default=30)
    ap.add_argument('--res', type=int, default=65)
    ap.add_argument('--precision', default='bf16-mixed')
    a = ap.parse_args()
    from ppsurf_amd import runner
    from ppsurf_amd.synthetic import write_dataset
    from test_gpu_cli import BASE, PPS, OPT
    tmp = tempfile.mkdtemp(prefix='pps_demo_')
    os.chdir(tmp)
    in_file = write_dataset(os.path.join(tmp, 'ds'), n_shapes=a.shapes, n_pts=12000, n_query=2000)
    cfg = dict(BASE); cfg.update(OPT)
    files = []
    for name, c in (('poco', cfg), ('pps', PPS), ('run', {'model': {'init_args': {'name': 'demo', 'gen_resolution_global': a.res,
                                                                                  'rec_batch_size': 50000, 'gen_refine_iter': 10}},
                                                          'data': {'init_args': {'in_file': in_file, 'batch_size': 10, 'manifold_points': 10000}},
                                                          'trainer': {'max_epochs': a.epochs, 'precision': a.precision},
                                                          'lr_scheduler': {'init_args': {'milestones': [int(a.epochs * 0.6), int(a.epochs * 0.85)]}}})):
        files += ['-c', os.path.join(tmp, name + '.yaml')]
        yaml.safe_dump(c, open(files[-1], 'w'))
    t0 = time.time()
    runner.main(['pps.py', 'fit'] + files)
    t_fit = time.time() - t0
    recs = [json.loads(l) for l in open(os.path.join(tmp, 'models', 'demo', 'version_0', 'metrics.jsonl'))]
    steps = [r for r in recs if 'step' in r]
    vals = [r for r in recs if 'loss/val/00_all' in r]
    print('fit: {} steps in {:.1f} s ({:.1f} ms/step incl. validation); train loss {:.3f} -> {:.3f}, accuracy {:.3f} -> {:.3f}; val loss {:.3f} -> {:.3f}'.format(
        len(steps), t_fit, t_fit / len(steps) * 1e3, steps[0]['loss/train/00_all'], np.mean([s['loss/train/00_all'] for s in steps[-4:]]),
        steps[0]['metrics/train/accuracy'], np.mean([s['metrics/train/accuracy'] for s in steps[-4:]]), vals[0]['loss/val/00_all'], vals[-1]['loss/val/00_all']))
    ckpt = os.path.join(tmp, 'models', 'demo', 'version_0', 'checkpoints', 'last.ckpt')
    os.environ['PPS_VERBOSE'] = ''
    # reconstruct the first 3 shapes with the trained weights
    with open(os.path.join(tmp, 'ds', 'testset.txt'), 'w') as f:
        f.write('\n'.join('synth_{:03d}'.format(i) for i in range(3)) + '\n')
    t0 = time.time()
    runner.main(['pps.py', 'predict'] + files + ['--ckpt_path', ckpt, '--model.init_args.results_dir', os.path.join(tmp, 'res')])
    mesh_dir = os.path.join(tmp, 'res', 'demo', 'ds', 'meshes')
    meshes = sorted(os.listdir(mesh_dir)) if os.path.isdir(mesh_dir) else []
    print('predict: {} meshes in {:.1f} s: {}'.format(len(meshes), time.time() - t0, meshes))
    from ppsurf_amd import meshio
    for m in meshes:
        v = meshio.read_ply_vertices(os.path.join(mesh_dir, m))[:, :3]
        cloud = meshio.read_ply_vertices(os.path.join(tmp, 'ds', '04_pts_vis', m))[:, :3]
        # distance of every input point to the nearest mesh vertex (the mesh should pass through the cloud)
        d = np.sqrt(((cloud[::20, None, :] - v[None, ::4, :]) ** 2).sum(-1)).min(axis=1)
        print('  {}: {} mesh vertices; input points -> nearest mesh vertex: median {:.4f}, 95 % {:.4f} (grid step {:.4f})'.format(
            m, len(v), np.median(d), np.quantile(d, 0.95), 1.0 / (a.res - 1)))


if __name__ == '__main__':
    main()
