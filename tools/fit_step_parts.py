"""What bounds the replayed fit step of bench.py: the recorded graph, or the loader thread that builds the next batch beside it?
(a) the bench leg (graph replay + loader thread on a side stream), (b) the graph replayed on ONE prepared batch (no batch preparation at all),
(c) the batch preparation alone (patches, support sampling, id tables, CSR extras), (d) (c) inline before every replay.   python tools/fit_step_parts.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_workloads as workloads

N = 40


def timed(f, n=N):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


step = workloads.FitStep(batch=10, precision='bf16-mixed', graph=True)
for _ in range(8):
    step()
print('(a) replay + loader thread: {:.2f} / {:.2f} ms per step'.format(timed(step), timed(step)))
step.close()
step.fut = None
batch = step._prepare(0)
torch.cuda.synchronize()
counter = [100]


def replay_only():
    counter[0] += 1
    step.stepper.run(batch, counter[0])


replay_only()
print('(b) replay of one prepared batch: {:.2f} / {:.2f} ms per step'.format(timed(replay_only), timed(replay_only)))
print('(c) batch preparation alone: {:.2f} / {:.2f} ms per batch'.format(timed(lambda: step._prepare(1)), timed(lambda: step._prepare(0))))


def inline():
    counter[0] += 1
    step.stepper.run(step._prepare(counter[0]), counter[0])


print('(d) preparation inline + replay: {:.2f} ms per step'.format(timed(inline)))

# (e) the recorded graph replayed back to back: no wait for the previous replay, no refill of the static inputs -- what the host round trip between
# two steps costs
sig = step.stepper.signature(batch)
static, graph, logged, state = step.stepper.graphs[sig]
print('(e) graph.replay() back to back: {:.2f} / {:.2f} ms per step'.format(timed(graph.replay), timed(graph.replay)))
