"""Fit step (BASELINE config 3, bf16-mixed) eager vs replayed HIP graph.   python tools/time_fit_graph.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppsurf_amd import workloads

for graph in (False, True):
    fit = workloads.FitStep(batch=10, precision='bf16-mixed', graph=graph)
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
    del fit
    torch.cuda.empty_cache()
