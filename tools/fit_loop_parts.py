"""Where the real `pps.py fit` loop spends its time per step: the loader alone (host items + collate_on_device), and the loop of ppsurf_amd.fit.
Usage: python tools/fit_loop_parts.py"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppsurf_amd import data
from ppsurf_amd.synthetic import write_dataset

tmp = tempfile.mkdtemp()
in_file = write_dataset(os.path.join(tmp, 'ds'), n_shapes=100, n_pts=25000, n_query=2000)
ds = data.PPSurfDataset(in_file, 50, 0.05, 42, False, 10000, 2000, True)
t0 = time.perf_counter()
items = [ds[i] for i in range(10)]
t_items = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
for _ in range(3):
    b = ds.collate_on_device(items, 'cuda:0')
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    b = ds.collate_on_device(items, 'cuda:0')
torch.cuda.synchronize()
t_collate = (time.perf_counter() - t0) * 100
loader = data.DeviceBatchLoader(ds, 10, True, 'cuda:0')
n = 0
t0 = time.perf_counter()
for b in loader:
    n += 1
torch.cuda.synchronize()
t_loader = (time.perf_counter() - t0) / n * 1e3
print('10 host items {:.1f} ms | collate_on_device {:.1f} ms per batch | loader alone {:.1f} ms per batch ({} batches)'.format(t_items, t_collate, t_loader, n))
