"""Inside the real fit: is the 16-bit neighbour_max equal to the fp32 op between two casts (forward and input gradient) on the tensors the fit
hands it?   python tools/dbg/nmax_infit.py"""
import os
import shutil
import sys
import tempfile

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
import test_gpu_configs as T          # noqa: E402
from ppsurf_amd import runner, train_ops         # noqa: E402

new = train_ops.neighbour_max
stats = {'calls': 0, 'fwd_bad': 0, 'bwd_calls': 0, 'bwd_bad': 0, 'seen': set()}


def old(x, idx):
    out = train_ops._NeighbourMax.apply(x.float(), idx)
    return out.to(x.dtype) if x.dtype in train_ops.LOW else out


def wrapped(x, idx):
    out = new(x, idx)
    stats['calls'] += 1
    stats['seen'].add((str(x.dtype), tuple(x.shape), tuple(idx.shape), x.is_contiguous(), torch.is_autocast_enabled()))
    with torch.enable_grad():
        xo = x.detach().clone().requires_grad_(True)
        oo = old(xo, idx)
        xn = x.detach().clone().requires_grad_(True)
        on = new(xn, idx)
    if not (torch.equal(out, oo) and out.dtype == oo.dtype):
        stats['fwd_bad'] += 1
    if out.requires_grad:
        def hook(g):
            stats['bwd_calls'] += 1
            go, = torch.autograd.grad(oo, xo, g)
            gn, = torch.autograd.grad(on, xn, g)
            if not (torch.equal(go, gn) and go.dtype == gn.dtype):
                stats['bwd_bad'] += 1
                stats['seen'].add(('grad', str(g.dtype), g.is_contiguous(), float((go.float() - gn.float()).abs().max())))
        out.register_hook(hook)
    return out


train_ops.neighbour_max = wrapped
root = tempfile.mkdtemp()
shutil.copytree(os.path.join(T.GOLDEN, 'abc_mini4'), os.path.join(root, 'abc'))
os.chdir(root)
os.environ['PPS_FIT_GRAPH'] = '0'
runner.main(['pps.py', 'fit'] + T._stack('poco', 'ppsurf', 'ppsurf_mini') + [
    '--data.init_args.in_file', os.path.join(root, 'abc', 'testset.txt'), '--data.init_args.batch_size', '3',
    '--data.init_args.manifold_points', '5000', '--trainer.max_epochs', '2', '--trainer.check_val_every_n_epoch', '15',
    '--trainer.precision', 'bf16-mixed'])
print('NMAX', {k: v for k, v in stats.items() if k != 'seen'})
for s in sorted(stats['seen'], key=str):
    print('NMAX', s)
