import os, subprocess, sys
S = sys.argv[1] if len(sys.argv) > 1 else None
if S is None:
    for s in ('plain', 'opt', 'sidewarm', 'nowarm', 'grads_exist', 'global', 'nobackward', 'backward_nocapture_leaf'):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), s], capture_output=True, text=True, timeout=300)
        print('==', s, 'rc', r.returncode, '|', (r.stdout.strip().splitlines() or ['-'])[-1][:100], flush=True)
    sys.exit(0)
import torch
from torch import nn
net = nn.Sequential(nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 2)).cuda()
x = torch.randn(256, 64, device='cuda'); y = torch.randint(0, 2, (256,), device='cuda')
opt = torch.optim.AdamW(net.parameters(), lr=1e-3, capturable=True, fused=True)
def step(do_opt):
    for p in net.parameters(): p.grad = None
    nn.functional.cross_entropy(net(x), y).backward()
    if do_opt: opt.step()
if S == 'sidewarm':
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step(False)
    torch.cuda.current_stream().wait_stream(s)
elif S != 'nowarm':
    for _ in range(3): step(S == 'opt')
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
mode = 'global' if S == 'global' else 'thread_local'
if S == 'grads_exist':
    step(False)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, capture_error_mode=mode):
        nn.functional.cross_entropy(net(x), y).backward()
elif S == 'nobackward':
    with torch.cuda.graph(g, capture_error_mode=mode):
        out = nn.functional.cross_entropy(net(x), y)
elif S == 'backward_nocapture_leaf':
    xx = x.clone().requires_grad_(True)
    w = torch.randn(64, 64, device='cuda')
    with torch.cuda.graph(g, capture_error_mode=mode):
        xx.grad = None
        (xx @ w).sum().backward()
else:
    with torch.cuda.graph(g, capture_error_mode=mode):
        step(S == 'opt')
print('captured', flush=True)
g.replay(); torch.cuda.synchronize()
print('ok')
