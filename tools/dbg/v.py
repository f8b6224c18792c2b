import os, subprocess, sys
S = sys.argv[1] if len(sys.argv) > 1 else None
if S is None:
    for s in ('keep', 'del', 'keep_samestream', 'cut2_samestream', 'cut2_del'):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), s], capture_output=True, text=True, timeout=300)
        warn = 'WARN' if 'AccumulateGrad' in r.stderr else 'nowarn'
        print('==', s, 'rc', r.returncode, warn, '|', ' / '.join((r.stdout.strip().splitlines() or ['-'])[-2:])[:160], flush=True)
    sys.exit(0)
import torch
from torch import nn
net = nn.Sequential(nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 2)).cuda()
x = torch.randn(256, 64, device='cuda'); y = torch.randint(0, 2, (256,), device='cuda')
cut = S.startswith('cut2')
def fwd():
    h1 = net[1](net[0](x)); h1c = h1.detach().requires_grad_(True) if cut else h1
    return h1, h1c, nn.functional.cross_entropy(net[4](net[3](net[2](h1c))), y)
same = 'samestream' in S
s = torch.cuda.Stream() if same else torch.cuda.current_stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        for p in net.parameters(): p.grad = None
        h1, h1c, loss = fwd(); loss.backward()
        if cut: h1.backward(h1c.grad)
torch.cuda.synchronize()
if 'del' in S:
    del h1, h1c, loss
for p in net.parameters(): p.grad = None
kw = {'stream': s} if same else {}
g0 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g0, capture_error_mode='thread_local', **kw):
    h1, h1c, loss = fwd(); loss.backward()
    gr = h1c.grad if cut else None
print('captured g0', flush=True)
if cut:
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, pool=g0.pool(), capture_error_mode='thread_local', **kw):
        h1.backward(gr)
    print('captured g1', flush=True)
    g0.replay(); g1.replay()
else:
    g0.replay()
torch.cuda.synchronize()
print('ok', float(net[0].weight.grad.abs().sum()))
