"""debug: which part of the staged capture crashes (each scenario in its own process)."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SCEN = sys.argv[1] if len(sys.argv) > 1 else None
if SCEN is None:
    for s in ('mlp', 'mlp_pool', 'net_single', 'net_stage0_only', 'net_all_stages_one_graph', 'net_nopack', 'net_staged'):
        r = subprocess.run([sys.executable] + (['-X', 'faulthandler'] if os.environ.get('FH') else []) + [os.path.abspath(__file__), s], capture_output=True, text=True, timeout=600)
        tail = (r.stdout + r.stderr).strip().splitlines()
        print('==', s, 'rc', r.returncode, '|', ' / '.join(t[:150] for t in tail[-3:]), flush=True)
    sys.exit(0)
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch
from torch import nn
if SCEN.startswith('mlp'):
    net = nn.Sequential(nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 2)).cuda()
    x = torch.randn(256, 64, device='cuda'); y = torch.randint(0, 2, (256,), device='cuda')
    def fwd():
        h1 = net[1](net[0](x)); h1c = h1.detach().requires_grad_(True)
        loss = nn.functional.cross_entropy(net[4](net[3](net[2](h1c))), y)
        return h1, h1c, loss
    for _ in range(2):
        h1, h1c, loss = fwd(); loss.backward(); h1.backward(h1c.grad)
        for p in net.parameters(): p.grad = None
    torch.cuda.synchronize()
    g0, g1 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g0, capture_error_mode='thread_local'):
        h1, h1c, loss = fwd(); loss.backward()
    with torch.cuda.graph(g1, pool=g0.pool(), capture_error_mode='thread_local'):
        h1.backward(h1c.grad)
    g0.replay(); g1.replay(); torch.cuda.synchronize()
    print('ok', float(net[0].weight.grad.abs().sum()))
    sys.exit(0)
import bench_workloads as workloads
from ppsurf_amd import fit, sharding, train_graph as tg, optim
fs = workloads.FitStep(batch=2, n=2000, q=200, p=50, precision='bf16-mixed', graph=False, overlap_prep=False, n_batches=1)
batch = fs._prepare(0)
net = fs.net
for m in net.modules():
    if isinstance(m, nn.Dropout): m.p = 0.0
ctx = torch.autocast('cuda', dtype=torch.bfloat16)
def loss_of(b):
    with ctx:
        return nn.functional.cross_entropy(net.forward(b).float(), b['occ'], reduction='none').mean()
buckets = sharding.GradBuckets([p for p in net.parameters() if p.requires_grad], defer=True, groups=tg.parameter_stages(net))
def eager(staged):
    buckets.zero()
    if staged:
        with tg.staged() as st:
            st.backward(loss_of(batch))
    else:
        loss_of(batch).backward()
    buckets.pack_all(); buckets.finish(); tg.release_step_caches()
for _ in range(3): eager(SCEN != 'net_single')
torch.cuda.synchronize()
static = {k: v.clone() for k, v in batch.items() if torch.is_tensor(v)}
gs = [torch.cuda.CUDAGraph() for _ in range(3)]
if SCEN == 'net_single':
    with torch.cuda.graph(gs[0], capture_error_mode='thread_local'):
        buckets.zero(); loss_of(static).backward(); buckets.pack_all()
elif SCEN == 'net_all_stages_one_graph':
    with tg.staged() as st:
        with torch.cuda.graph(gs[0], capture_error_mode='thread_local'):
            buckets.zero(); st.backward(loss_of(static)); buckets.pack_all()
elif SCEN == 'net_stage0_only':
    with tg.staged() as st:
        with torch.cuda.graph(gs[0], capture_error_mode='thread_local'):
            buckets.zero(); st.run_stage(0, loss_of(static)); buckets.pack(0)
elif SCEN == 'net_nopack':
    with tg.staged() as st:
        with torch.cuda.graph(gs[0], capture_error_mode='thread_local'):
            buckets.zero(); st.run_stage(0, loss_of(static))
        for k in (1, 2):
            with torch.cuda.graph(gs[k], pool=gs[0].pool(), capture_error_mode='thread_local'):
                st.run_stage(k)
else:
    with tg.staged() as st:
        with torch.cuda.graph(gs[0], capture_error_mode='thread_local'):
            buckets.zero(); st.run_stage(0, loss_of(static)); buckets.pack(0)
        for k in (1, 2):
            with torch.cuda.graph(gs[k], pool=gs[0].pool(), capture_error_mode='thread_local'):
                st.run_stage(k); buckets.pack(k)
print('captured')
for g in gs:
    try: g.replay()
    except Exception as e: print('replay skipped', str(e)[:60])
torch.cuda.synchronize()
print('ok')
