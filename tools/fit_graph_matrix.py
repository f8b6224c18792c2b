"""Debug aid: which batch geometries survive capture + replay of the fit step (each in its own process)."""
import subprocess, sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, torch
sys.path.insert(0, %r)
from ppsurf_amd import workloads
b, n, q, prec = %d, %d, %d, %r
fit = workloads.FitStep(batch=b, n=n, q=q, precision=prec, graph=True)
for i in range(8):
    l = fit(); torch.cuda.synchronize()
print("OK", float(l), len(fit.stepper.graphs), fit.stepper.failed)
'''
for cfg in [(4, 2000, 300, '32'), (10, 2000, 300, 'bf16-mixed'), (4, 10000, 300, 'bf16-mixed'), (4, 2000, 2000, 'bf16-mixed'), (10, 10000, 2000, '32'),
            (10, 10000, 2000, 'bf16-mixed')]:
    r = subprocess.run([sys.executable, '-c', code % ((REPO,) + cfg)], capture_output=True, text=True)
    tail = [l for l in (r.stdout + r.stderr).split('\n') if l.strip() and 'amdgpu' not in l][-2:]
    print(cfg, 'rc', r.returncode, tail, flush=True)
