"""Fit step (BASELINE config 3, bf16-mixed) eager vs replayed HIP graph, each in its own process.   python tools/time_fit_graph.py"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, time, torch
sys.path.insert(0, %r)
from ppsurf_amd import workloads
graph = %r
fit = workloads.FitStep(batch=10, precision='bf16-mixed', graph=True)
fit.stepper.enabled = graph
for _ in range(6):
    loss = fit()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    loss = fit()
torch.cuda.synchronize()
print('graph' if graph else 'eager', '{:.2f} ms/step, loss {:.5f}, graphs captured: {}, failed: {}'.format(
    (time.perf_counter() - t0) / n * 1e3, float(loss), len(fit.stepper.graphs), fit.stepper.failed))
'''
for graph in (False, True):
    r = subprocess.run([sys.executable, '-c', code % (REPO, graph)], capture_output=True, text=True)
    print('\n'.join(l for l in (r.stdout + r.stderr).split('\n') if l.strip() and 'amdgpu' not in l)[-400:], flush=True)
