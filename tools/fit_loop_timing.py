import os, sys, tempfile, time, json, yaml
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from ppsurf_amd import runner
from ppsurf_amd.synthetic import write_dataset
from test_gpu_cli import BASE, PPS, OPT
tmp = tempfile.mkdtemp(); os.chdir(tmp)
NS = int(os.environ.get('SHAPES', 100)); NE = int(os.environ.get('EPOCHS', 4))
in_file = write_dataset(os.path.join(tmp, 'ds'), n_shapes=NS, n_pts=25000, n_query=2000)
cfg = dict(BASE); cfg.update(OPT)
files = []
for name, c in (('poco', cfg), ('pps', PPS), ('run', {'model': {'init_args': {'name': 'demo'}},
        'data': {'init_args': {'in_file': in_file, 'batch_size': 10, 'manifold_points': 10000}},
        'trainer': {'max_epochs': NE, 'precision': os.environ.get('PRECISION', 'bf16-mixed'), 'check_val_every_n_epoch': 0}})):
    files += ['-c', os.path.join(tmp, name + '.yaml')]; yaml.safe_dump(c, open(files[-1], 'w'))
t0 = time.time(); runner.main(['pps.py', 'fit'] + files); dt = time.time() - t0
steps = NS // 10 * NE
print('fit loop: %d steps in %.2f s -> %.1f ms/step (first epoch includes warm-up)' % (steps, dt, dt / steps * 1e3))
