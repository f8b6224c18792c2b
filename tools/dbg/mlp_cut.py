import os, subprocess, sys
S = sys.argv[1] if len(sys.argv) > 1 else None
if S is None:
    for s in ('nocut', 'cut_full_in_g0', 'cut_two_graphs', 'cut_two_graphs_nopool', 'cut_keepgrad'):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), s], capture_output=True, text=True, timeout=300)
        print('==', s, 'rc', r.returncode, '|', ' / '.join((r.stdout.strip().splitlines() or ['-'])[-3:])[:160], flush=True)
    sys.exit(0)
import torch
from torch import nn
net = nn.Sequential(nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 2)).cuda()
x = torch.randn(256, 64, device='cuda'); y = torch.randint(0, 2, (256,), device='cuda')
def fwd(cut):
    h1 = net[1](net[0](x)); h1c = h1.detach().requires_grad_(True) if cut else h1
    loss = nn.functional.cross_entropy(net[4](net[3](net[2](h1c))), y)
    return h1, h1c, loss
cut = S != 'nocut'
for _ in range(2):
    for p in net.parameters(): p.grad = None
    h1, h1c, loss = fwd(cut); loss.backward()
    if cut: h1.backward(h1c.grad)
torch.cuda.synchronize()
print('warm', flush=True)
for p in net.parameters(): p.grad = None
g0 = torch.cuda.CUDAGraph()
if S in ('nocut', 'cut_full_in_g0'):
    with torch.cuda.graph(g0, capture_error_mode='thread_local'):
        h1, h1c, loss = fwd(cut); loss.backward()
        if cut: h1.backward(h1c.grad)
    print('captured', flush=True)
    g0.replay()
else:
    with torch.cuda.graph(g0, capture_error_mode='thread_local'):
        h1, h1c, loss = fwd(True); loss.backward()
        gr = h1c.grad
    print('captured g0', flush=True)
    g1 = torch.cuda.CUDAGraph()
    kw = {} if S == "cut_two_graphs_nopool" else {"pool": g0.pool()}
    with torch.cuda.graph(g1, capture_error_mode='thread_local', **kw):
        h1.backward(gr)
    print('captured g1', flush=True)
    g0.replay(); g1.replay()
torch.cuda.synchronize()
print('ok', float(net[0].weight.grad.abs().sum()))
