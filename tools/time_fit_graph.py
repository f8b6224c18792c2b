"""The fit step exactly as bench.py's `fit_ms_per_step` leg runs it (workloads.FitStep(graph=True): replayed HIP graph, batch assembly on the
loader thread / side stream), alone.  Usage: python tools/time_fit_graph.py [--steps 60] [--precision bf16-mixed|16-mixed] [--eager]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_workloads as workloads          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--precision', default='bf16-mixed')
    ap.add_argument('--eager', action='store_true')
    a = ap.parse_args()
    step = workloads.FitStep(batch=10, precision=a.precision, graph=not a.eager)
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    from ppsurf_amd.fit import HostGcPacer
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    print('gpu state before:', bench.gpu_state())
    # the collector paced like in ppsurf_amd.fit / bench.py's fit leg: without it a full collection (60-100 ms with everything a fit keeps alive)
    # falls into every second or third block of 40-60 steps and reads as "the step got 1-2 ms slower after a while" (round 6 fell for it once)
    with HostGcPacer() as pacer:
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(a.steps):
                loss = step()
                pacer.tick()
            torch.cuda.synchronize()
            print('{} {}: {:.2f} ms/step  loss {:.4f}'.format(a.precision, 'eager' if a.eager else 'graph', (time.perf_counter() - t0) / a.steps * 1e3, float(loss)))
    print('gpu state after:', bench.gpu_state())
    step.close()


if __name__ == '__main__':
    main()
