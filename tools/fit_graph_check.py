import sys, torch
sys.path.insert(0, '.')
import bench_workloads as workloads
import random
res = {}
for graph in (False, True):
    random.seed(0); torch.manual_seed(0)
    B, N, Q = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 4), (3, 2000), (4, 300)))        # python tools/fit_graph_check.py [precision] [B N Q]
    fit = workloads.FitStep(batch=B, n=N, q=Q, precision=(sys.argv[1] if len(sys.argv) > 1 else '32'), graph=True, n_batches=2)
    fit.stepper.enabled = graph                            # same fused / capturable AdamW in both runs
    for m in fit.net.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
    losses = []
    for i in range(8):
        random.seed(100 + i); torch.manual_seed(100 + i)          # the support sampling draws from both
        losses.append(float(fit()))
    res[graph] = (losses, {k: v.clone() for k, v in fit.net.state_dict().items()})
    print('graph' if graph else 'eager', ['%.6f' % l for l in losses], len(fit.stepper.graphs))
a, b = res[False], res[True]
print('max loss diff', max(abs(x - y) for x, y in zip(a[0], b[0])))
d = sorted(((float((a[1][k].float() - b[1][k].float()).abs().max()), k) for k in a[1]), reverse=True)
print('largest state-dict differences', d[:8])
import hashlib
for g in (False, True):
    h = hashlib.sha1()
    for k in sorted(res[g][1]):
        h.update(res[g][1][k].float().cpu().numpy().tobytes())
    print('digest', 'graph' if g else 'eager', h.hexdigest()[:16])
print('tensors that differ: {} of {}'.format(sum(1 for x in d if x[0] > 0), len(d)))
