"""Does the fit step's time depend on what the process allocated before it, and is torch.cuda.empty_cache() the reason?  (profiles/NOTES_r6.md section 2)
    python tools/dbg/fit_after_history.py [--empty 0|1] [--history 0|1]
fresh: FitStep timed first; then (history) the default bench's inference legs (205 chunks over 5 shapes, the config-5 leg), freed, optionally
empty_cache(), then a NEW FitStep timed the same way."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                   # noqa: E402
import bench_workloads as workloads            # noqa: E402
from ppsurf_amd.fit import HostGcPacer         # noqa: E402


def timed_fit(tag):
    fit = workloads.FitStep(batch=10, precision='bf16-mixed', device='cuda:0', graph=True)
    for _ in range(8):
        fit()
    torch.cuda.synchronize()
    out = []
    with HostGcPacer() as pacer:
        for _ in range(2):
            t0 = time.perf_counter()
            for _ in range(60):
                fit()
                pacer.tick()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / 60 * 1e3)
    fit.close()
    del fit
    print('{}: {:.2f} / {:.2f} ms per step; allocated {:.2f} GB reserved {:.2f} GB'.format(tag, out[0], out[1], torch.cuda.memory_allocated() / 1e9,
                                                                                        torch.cuda.memory_reserved() / 1e9), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--empty', type=int, default=1)
    ap.add_argument('--history', type=int, default=1)
    ap.add_argument('--fresh-first', type=int, default=1)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    if a.fresh_first:
        timed_fit('fresh process')
        if a.empty:
            torch.cuda.empty_cache()
    if a.history:
        from ppsurf_amd.decoder import DecoderPlan
        from ppsurf_amd.synthetic import network_state_dict
        plan = DecoderPlan(network_state_dict('ppsurf'), dev, dtype='f16x3')
        shapes, work = bench.build_work(plan, 205, 0, dev)
        for pipe, c in work[:100]:
            pipe.run([c])
        torch.cuda.synchronize()
        del shapes, work, plan
        if a.empty:
            torch.cuda.empty_cache()
        bench.config5_leg(dev, 'f16x3')
        if a.empty:
            torch.cuda.empty_cache()
        print('history done; reserved {:.2f} GB'.format(torch.cuda.memory_reserved() / 1e9), flush=True)
    timed_fit('after history, empty_cache={}'.format(a.empty))


if __name__ == '__main__':
    main()
