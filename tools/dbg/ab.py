import os, subprocess, sys
S = sys.argv[1] if len(sys.argv) > 1 else None
if S is None:
    for s in ('3_in', '3_out', '5_in', '5_out', '5_out_opt', '5_out_warm3'):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), s], capture_output=True, text=True, timeout=300)
        print('==', s, 'rc', r.returncode, '|', ' / '.join((r.stdout.strip().splitlines() or ['-'])[-2:])[:160], flush=True)
    sys.exit(0)
import torch
from torch import nn
layers = [nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 2)] if S.startswith('3') else [nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 2)]
net = nn.Sequential(*layers).cuda()
x = torch.randn(256, 64, device='cuda'); y = torch.randint(0, 2, (256,), device='cuda')
if 'opt' in S:
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, capturable=True, fused=True)
for _ in range(3 if 'warm3' in S else 2):
    for p in net.parameters(): p.grad = None
    nn.functional.cross_entropy(net(x), y).backward()
torch.cuda.synchronize()
print('warm', flush=True)
g = torch.cuda.CUDAGraph()
if '_out' in S:
    for p in net.parameters(): p.grad = None
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    if '_in' in S:
        for p in net.parameters(): p.grad = None
    nn.functional.cross_entropy(net(x), y).backward()
print('captured', flush=True)
g.replay(); torch.cuda.synchronize()
print('ok')
