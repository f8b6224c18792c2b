"""Debug aid: sequences of FitStep phases in one process (each variant in its own subprocess)."""
import subprocess, sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, torch, os, gc
sys.path.insert(0, %r)
import bench_workloads as workloads
seq = %r
for tok in seq.split():
    if tok == "empty":
        gc.collect(); torch.cuda.empty_cache(); continue
    graph, fused = tok.startswith("g"), tok != "e"
    fit = workloads.FitStep(batch=10, n=10000, q=2000, precision="bf16-mixed", graph=(tok != "e"))
    fit.stepper.enabled = graph
    for i in range(int(os.environ.get("NSTEPS", 10))):
        l = fit()
        if os.environ.get("TRACE"):
            torch.cuda.synchronize(); print(tok, "step", i, float(l), torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, flush=True)
    torch.cuda.synchronize()
    print(tok, "ok", float(l), len(fit.stepper.graphs), flush=True)
    del fit
print("DONE")
'''
for seq in sys.argv[1:] or ['e g', 'e empty g', 'E empty g']:
    r = subprocess.run([sys.executable, '-c', code % (REPO, seq)], capture_output=True, text=True)
    lines = [l for l in (r.stdout + r.stderr).split('\n') if l.strip() and 'amdgpu' not in l]
    print(repr(seq), 'rc', r.returncode, [l for l in lines if ' ok ' in l or 'DONE' in l or 'Kernel Name' in l or 'error' in l.lower() or ' step ' in l][-8:], flush=True)
