"""Is the short fit of tests/test_gpu_configs.py's `trained` fixture the same from run to run / between builds?  Prints a digest of the checkpoint's
state_dict after each of N fits in fresh directories.   python tools/dbg/fit_repro.py [runs=2] [epochs=30]"""
import hashlib
import os
import shutil
import sys
import tempfile

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import test_gpu_configs as T          # noqa: E402
from ppsurf_amd import runner         # noqa: E402

if os.environ.get('PPS_DBG_NODROPOUT'):
    import torch.nn.functional as _F
    _F.dropout = lambda x, p=0.5, training=True, inplace=False: x
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
epochs = sys.argv[2] if len(sys.argv) > 2 else '30'


def poison(value, gb=40):
    """fills the caching allocator's free blocks with a pattern: whatever the fit then reads without having written it is that pattern"""
    blocks = []
    for size in (1 << 30, 1 << 27, 1 << 24, 1 << 21, 1 << 18, 1 << 15, 1 << 12, 1 << 9):
        for _ in range(max(4, min(256, (gb << 30) // size // 8))):
            blocks.append(torch.full((size // 4,), value, dtype=torch.float32, device='cuda'))
    torch.cuda.synchronize()
    del blocks


for r in range(runs):
    if os.environ.get('PPS_DBG_POISON'):
        torch.cuda.empty_cache()
        poison(float(os.environ['PPS_DBG_POISON']))
    root = tempfile.mkdtemp()
    shutil.copytree(os.path.join(T.GOLDEN, 'abc_mini4'), os.path.join(root, 'abc'))
    cwd = os.getcwd()
    os.chdir(root)
    try:
        runner.main(['pps.py', 'fit'] + T._stack('poco', 'ppsurf', 'ppsurf_mini') + [
            '--data.init_args.in_file', os.path.join(root, 'abc', 'testset.txt'), '--data.init_args.batch_size', '3',
            '--data.init_args.manifold_points', '5000', '--trainer.max_epochs', epochs, '--trainer.check_val_every_n_epoch', '15',
            '--trainer.precision', 'bf16-mixed', '--lr_scheduler.init_args.milestones', '[22, 27]'])
    finally:
        os.chdir(cwd)
    sd = torch.load(os.path.join(root, 'models', 'ppsurf_mini', 'version_0', 'checkpoints', 'last.ckpt'), map_location='cpu')['state_dict']
    h = hashlib.sha1()
    for k in sorted(sd):
        h.update(sd[k].contiguous().numpy().tobytes() if sd[k].dtype != torch.bfloat16 else sd[k].float().numpy().tobytes())
    print('FIT', r, h.hexdigest()[:16], flush=True)
    if os.environ.get('PPS_DBG_SAVE'):
        torch.save({k: v.float() for k, v in sd.items()}, os.environ['PPS_DBG_SAVE'])
    if os.environ.get('PPS_DBG_CMP'):
        other = torch.load(os.environ['PPS_DBG_CMP'])
        d = sorted(((float((other[k] - sd[k].float()).abs().max()), float(other[k].abs().max()), k) for k in sd), reverse=True)
        print('DIFF tensors that differ: {} of {}'.format(sum(1 for x in d if x[0] > 0), len(d)))
        for x in d[:25]:
            print('DIFF {:.3e} (max |value| {:.3e}) {}'.format(*x))
        same = [x[2] for x in d if x[0] == 0]
        print('DIFF equal:', same[:40])
    shutil.rmtree(root, ignore_errors=True)
