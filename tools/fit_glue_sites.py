"""Which lines of the fit step launch the small torch kernels (copies, casts, fills, cats, reductions) that sit between the HIP ops?
The EAGER step (same ops as the recorded graph) under a TorchDispatchMode: per (aten op, innermost ppsurf_amd / bench frame, shapes) the calls per
step -- forward, and the backward functions that run inside the autograd engine.    python tools/fit_glue_sites.py [--steps 2]"""
import argparse
import collections
import os
import sys

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_workloads as workloads          # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/'
SKIP = ('aten.view', 'aten._unsafe_view', 'aten.detach', 'aten.alias', 'aten.t.', 'aten.transpose', 'aten.permute', 'aten.expand', 'aten.slice', 'aten.select',
        'aten.unsqueeze', 'aten.squeeze', 'aten.as_strided', 'aten.empty', 'aten.reshape', 'aten.narrow', 'aten.split', 'aten.unbind', 'aten.lift_fresh',
        'aten.is_', 'aten.sym_', 'aten.stride', 'aten.size', 'aten.numel', 'aten._local_scalar_dense', 'aten.new_empty', 'aten.unfold', 'aten.chunk')


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            try:
                f = sys._getframe(1)
            except ValueError:          # called from the autograd engine's thread for a built-in backward node
                f = None
            site = '(autograd engine: built-in backward node)' if f is None else '?'
            while f is not None:
                fn = f.f_code.co_filename
                if ('/ppsurf_amd/' in fn or 'bench_workloads' in fn) and 'fit_glue_sites' not in fn:
                    site = '{}:{} {}'.format(fn.replace(ROOT, ''), f.f_lineno, f.f_code.co_name)
                    break
                f = f.f_back
            ts = [a for a in list(args) + list((kwargs or {}).values()) if torch.is_tensor(a)]
            if not ts:
                ts = [x for a in args if isinstance(a, (list, tuple)) for x in a if torch.is_tensor(x)][:1]
            if any(t.is_cuda for t in ts) or not ts:
                desc = ' '.join('{}{}'.format(str(t.dtype).replace('torch.', ''), list(t.shape)) for t in ts[:2])
                self.agg[(name, site, desc)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=2)
    a = ap.parse_args()
    step = workloads.FitStep(batch=10, precision='bf16-mixed', graph=False)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    mode = Sites()
    with mode:
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
    step.close()
    print('aten calls per step on CUDA tensors (main thread + autograd engine; the loader thread is not under the mode): {:.0f}'.format(
        sum(mode.agg.values()) / a.steps))
    for (name, site, desc), c in sorted(mode.agg.items(), key=lambda kv: (-kv[1], kv[0]))[:400]:
        print('{:5.1f} x  {:34s} {:70s} {}'.format(c / a.steps, name, site, desc))


if __name__ == '__main__':
    main()
