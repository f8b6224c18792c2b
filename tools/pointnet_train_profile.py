"""Kernel-level breakdown of the PointNet part of a fit step (forward + backward, bf16-mixed, 20000 queries x 50 points)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ppsurf_amd import modules, train_graph as tg
from ppsurf_amd.synthetic import network_state_dict

dev = torch.device('cuda')
net = modules.PPSurfNetwork(in_channels=3, latent_size=256, out_channels=2, k=64, num_pts_local=50, pointnet_latent_size=256)
net.load_state_dict(network_state_dict('ppsurf'))
net = net.to(dev).train()
patches = torch.randn(20000, 50, 3, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else 'pointnet'

def step():
    net.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        if which == 'pointnet':
            out = tg.pointnet(net.point_net, patches)[0]
        else:
            out = tg.mlp(net.mlp, torch.randn(20000, 256, device=dev))
    out.float().square().mean().backward()

for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    t = getattr(e, 'device_time_total', 0) or getattr(e, 'cuda_time_total', 0)
    if t > 0 and e.device_type.name != 'CPU':
        rows.append((t / 3e3, e.count // 3, e.key[:90]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('total device time per step {:.2f} ms'.format(tot))
for t, c, k in rows[:28]:
    print('{:8.3f} ms  x{:<4d} {}'.format(t, c, k))
