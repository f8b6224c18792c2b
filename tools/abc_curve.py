import os, sys, json, shutil, tempfile, yaml, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from golden_util import GOLDEN
from ppsurf_amd import runner
from test_gpu_cli import BASE, PPS, OPT
tmp = tempfile.mkdtemp(); os.chdir(tmp)
shutil.copytree(os.path.join(GOLDEN, 'abc_mini4'), os.path.join(tmp, 'abc'))
in_file = os.path.join(tmp, 'abc', 'testset.txt')
cfg = dict(BASE); cfg.update(OPT)
files = []
E = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, c in (('poco', cfg), ('pps', PPS), ('run', {'model': {'init_args': {'name': 'abc', 'gen_resolution_global': 49, 'rec_batch_size': 30000, 'gen_refine_iter': 5, 'gen_subsample_manifold': 5000}},
        'data': {'init_args': {'in_file': in_file, 'batch_size': 3, 'manifold_points': 5000}},
        'trainer': {'max_epochs': E, 'precision': 'bf16-mixed', 'check_val_every_n_epoch': 0},
        'lr_scheduler': {'init_args': {'milestones': [int(E*0.7), int(E*0.9)]}}})):
    files += ['-c', os.path.join(tmp, name + '.yaml')]; yaml.safe_dump(c, open(files[-1], 'w'))
runner.main(['pps.py', 'fit'] + files)
recs = [json.loads(l) for l in open(os.path.join(tmp, 'models', 'abc', 'version_0', 'metrics.jsonl'))]
steps = [r for r in recs if 'step' in r]
for i in range(0, len(steps), max(1, len(steps)//12)):
    print(steps[i]['step'], round(steps[i]['loss/train/00_all'], 3), round(steps[i]['metrics/train/accuracy'], 3))
print('last5', np.mean([s['metrics/train/accuracy'] for s in steps[-5:]]), np.mean([s['loss/train/00_all'] for s in steps[-5:]]))
