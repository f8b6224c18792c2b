"""Timeline statistics of the replayed fit step from a rocprofv3 kernel trace in CSV form
(`rocprofv3 --kernel-trace --output-format csv -d DIR -o trace -- python tools/time_fit_graph.py --steps 10`):
per step (delimited by the one `adamw_pieces_kernel` launch of a step) the span, the time during which at least one / at least two kernels run,
the sum of kernel durations per hardware queue, the gaps, and the kernels by total time.

    python tools/fit_timeline.py DIR [steps_from_the_end=10] [--by-grid] [--per-queue]  -> text on stdout   (--by-grid: launches of one kernel told apart by grid
    size; --per-queue: the kernel table once more per hardware queue -- the step's queue and the loader's)
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0][:64]


def union(intervals):
    """total length of the union and of the part covered at least twice"""
    ev = []
    for s, e in intervals:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    one = two = 0
    depth, last = 0, None
    for t, d in ev:
        if last is not None and depth >= 1:
            one += t - last
            if depth >= 2:
                two += t - last
        depth += d
        last = t
    return one, two


def main(src, nsteps, by_grid=False, per_queue=False):
    files = glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
    if not files:
        raise SystemExit('no *kernel_trace.csv under ' + src)
    rows = []
    with open(files[0]) as fh:
        for r in csv.DictReader(fh):
            name = short(r['Kernel_Name'])
            if by_grid and r.get('Grid_Size_X', r.get('Grid_Size')):
                name = '{} [grid {}]'.format(name[:50], r.get('Grid_Size_X', r.get('Grid_Size')))
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name, r.get('Queue_Id', '0'), r.get('Stream_Id', '0')))
    rows.sort()
    marks = [e for s, e, n, q, st in rows if n.startswith('adamw_pieces_kernel')]
    if len(marks) < nsteps + 1:
        raise SystemExit('only {} steps in the trace'.format(len(marks)))
    t0, t1 = marks[-nsteps - 1], marks[-1]
    win = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    span = (t1 - t0) / 1e3 / nsteps
    one, two = union([(s, e) for s, e, *_ in win])
    print('steps analysed: {}   kernels per step: {:.0f}'.format(nsteps, len(win) / nsteps))
    print('span per step            {:9.1f} us'.format(span))
    print('>= 1 kernel running      {:9.1f} us   ({:.1f} % of the span; idle {:.1f} us)'.format(one / 1e3 / nsteps, 100 * one / (t1 - t0), span - one / 1e3 / nsteps))
    print('>= 2 kernels running     {:9.1f} us'.format(two / 1e3 / nsteps))
    print('sum of kernel durations  {:9.1f} us'.format(sum(e - s for s, e, *_ in win) / 1e3 / nsteps))
    per_q = defaultdict(list)
    for s, e, n, q, st in win:
        per_q[(q, st)].append((s, e))
    print('per (queue, stream): kernels per step, sum of durations, busy time')
    for key, iv in sorted(per_q.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
        print('   queue {:>3s} stream {:>3s}: {:7.0f} {:9.1f} us {:9.1f} us'.format(key[0], key[1], len(iv) / nsteps, sum(e - s for s, e in iv) / 1e3 / nsteps,
                                                                                 union(iv)[0] / 1e3 / nsteps))
    # gaps of the union timeline
    iv = sorted((s, e) for s, e, *_ in win)
    gaps, cur = [], iv[0][1]
    for s, e in iv[1:]:
        if s > cur:
            gaps.append(s - cur)
        cur = max(cur, e)
    gaps.sort()
    if gaps:
        print('idle gaps per step: {:.0f}, total {:.1f} us, median {:.2f} us, p90 {:.2f} us, max {:.1f} us'.format(
            len(gaps) / nsteps, sum(gaps) / 1e3 / nsteps, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3))
    tot = defaultdict(lambda: [0, 0])
    for s, e, n, q, st in win:
        tot[n][0] += 1
        tot[n][1] += e - s
    print('{:64s} {:>8s} {:>10s} {:>9s}'.format('kernel', 'per step', 'us / step', 'avg us'))
    for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:(140 if by_grid else 70)]:
        print('{:64s} {:8.1f} {:10.1f} {:9.2f}'.format(n, c / nsteps, d / 1e3 / nsteps, d / 1e3 / c))
    if per_queue:
        for key in sorted(per_q, key=lambda k: -len(per_q[k])):
            tq = defaultdict(lambda: [0, 0])
            for s, e, n, q, st in win:
                if (q, st) == key:
                    tq[n][0] += 1
                    tq[n][1] += e - s
            print('--- queue {} stream {}: kernels by total time'.format(*key))
            for n, (c, d) in sorted(tq.items(), key=lambda kv: -kv[1][1])[:90]:
                print('{:64s} {:8.1f} {:10.1f} {:9.2f}'.format(n, c / nsteps, d / 1e3 / nsteps, d / 1e3 / c))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10, by_grid='--by-grid' in sys.argv, per_queue='--per-queue' in sys.argv)
