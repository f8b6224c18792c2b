"""Debug aid: which variants of the fit step survive capture + replay at full size, bf16-mixed (each in its own process)."""
import subprocess, sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, torch, os
sys.path.insert(0, %r)
from ppsurf_amd import train_graph, train_ops
import bench_workloads as workloads
knob = %r
if "nosplitk" in knob: train_graph.SPLITK_MIN_ROWS = 10**12
if "s64" in knob: train_graph.SPLITK_SLABS = 10**9
if "s256" in knob: train_graph.SPLITK_SLABS = 256
if "s32" in knob: train_graph.SPLITK_SLABS = 32
if "nobn" in knob: train_ops.bn_supported = lambda r, c: False
if "nocache" in knob:
    _ac = torch.autocast
    class AC(_ac):
        def __init__(self, *a, **k):
            k["cache_enabled"] = False
            super().__init__(*a, **k)
    torch.autocast = AC
fit = workloads.FitStep(batch=10, n=10000, q=2000, precision="bf16-mixed", graph=True)
if "noenc" in knob:
    fit.net.encoder.requires_grad_(False)
for i in range(int(os.environ.get('NSTEPS', 8))):
    l = fit()
    if 'sync' in knob: torch.cuda.synchronize()
torch.cuda.synchronize()
print("OK", float(l), len(fit.stepper.graphs), fit.stepper.failed)
'''
for knob in sys.argv[1:] or ['sync', 'nosync']:
    r = subprocess.run([sys.executable, '-c', code % (REPO, knob)], capture_output=True, text=True)
    tail = [l for l in (r.stdout + r.stderr).split('\n') if l.strip() and 'amdgpu' not in l and 'coredump' not in l and 'core dump' not in l][-1:]
    print(knob, 'rc', r.returncode, tail, flush=True)
