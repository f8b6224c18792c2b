"""Does the fit step read memory it did not write?  The eager config-3 step (small: B=3 x 5000 points x 500 queries) from the same start, with the
caching allocator's free blocks filled beforehand with (a) zeros, (b) NaN patterns, (c) large finite values: the parameter digests after 3 steps
must be equal.   python tools/dbg/fit_poison.py [B=3] [N=5000] [Q=500]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_workloads as workloads          # noqa: E402

B, N, Q = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 3), (2, 5000), (3, 500)))


def poison(value, gb=12):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    blocks = []
    for size in (1 << 30, 1 << 26, 1 << 22, 1 << 18, 1 << 14, 1 << 10):        # blocks of every size class the allocator hands out again
        for _ in range(max(1, min(64, (gb << 30) // size // 6))):
            blocks.append(torch.full((size // 4,), value, dtype=torch.float32, device='cuda'))
    torch.cuda.synchronize()
    del blocks                                   # back into the allocator's cache, contents intact


def digest(net, loss):
    h = hashlib.sha1()
    for k, v in sorted(net.state_dict().items()):
        h.update(v.detach().float().cpu().numpy().tobytes())
    return h.hexdigest()[:16], float(loss)


for name, value in (('zeros', 0.0), ('nan', float('nan')), ('1e30', 1e30), ('zeros again', 0.0)):
    torch.manual_seed(0)
    step = workloads.FitStep(batch=B, n=N, q=Q, precision=os.environ.get('PPS_DBG_PRECISION', 'bf16-mixed'), graph=False, overlap_prep=False)
    loss = None
    for _ in range(3):
        poison(value)
        loss = step()
    torch.cuda.synchronize()
    print('POISON {:12s} {} loss {:.6f}'.format(name, *digest(step.net, loss)), flush=True)
    step.close()
    del step
