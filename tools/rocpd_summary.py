"""Summarise rocprofv3 (rocpd sqlite) outputs of tools/profile_round.sh into small text/json files for profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_<tag> profiles/<prefix>
"""
import glob
import json
import os
import sqlite3
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0][:60]


def main(src, dst):
    out = {}
    lines = []
    tr = glob.glob(os.path.join(src, 'trace', '*.db'))
    if tr:
        c = sqlite3.connect(tr[0])
        lines.append('# rocprofv3 --kernel-trace --stats (top kernels; durations in us)')
        lines.append('{:60s} {:>6s} {:>12s} {:>11s} {:>7s}'.format('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
        for name, calls, total, avg, pct in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
            if pct < 0.005:
                continue
            lines.append('{:60s} {:6d} {:12.1f} {:11.2f} {:7.2f}'.format(short(name), calls, total / 1e3 if total > 1e7 else total, avg / 1e3 if avg > 1e6 else avg, pct))
            out.setdefault(short(name), {})['avg_us'] = avg
            out[short(name)]['calls'] = calls
    for sub in sorted(glob.glob(os.path.join(src, 'pmc_*'))):
        dbs = glob.glob(os.path.join(sub, '*.db'))
        if not os.path.isdir(sub) or not dbs:
            continue
        c = sqlite3.connect(dbs[0])
        lines.append('')
        lines.append('# rocprofv3 --pmc pass {} (average per dispatch)'.format(os.path.basename(sub)))
        q = 'select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name'
        for name, counter, n, avg in c.execute(q):
            if 'at::native' in name or 'rocclr' in name:
                continue
            lines.append('{:60s} {:32s} n={:3d} avg={:.6g}'.format(short(name), counter, n, avg))
            out.setdefault(short(name), {})[counter] = avg
            out[short(name)].setdefault('n', {})[counter] = n
    open(dst + '_rocprof_summary.txt', 'w').write('\n'.join(lines) + '\n')
    # the dominant kernel of the run: in an f16x3 run the fp32 kernel of the same name also appears (the gated fall-back launch, a few microseconds)
    k = max((out.get(n, {}) for n in ('interp_pool_kernel', 'interp_pool_f16x3_kernel')), key=lambda v: v.get('avg_us', 0.0))
    # the GPU box has no .git: the caller passes the commit the snapshot was taken at (PPS_GIT_HEAD=$(git rev-parse --short=12 HEAD))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    import bench_workloads
    pmc = {'source': os.path.basename(src.rstrip('/')), 'git_head': os.environ.get('PPS_GIT_HEAD', 'unrecorded'),
           'csrc_digest': bench_workloads.csrc_digest(), 'kernels': out}
    if 'FETCH_SIZE' in k and 'WRITE_SIZE' in k:
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide
        # coalesced read stream -> doubled; WRITE_SIZE is uncalibrated and taken as is.
        pmc['interp_pool_hbm_bytes_per_launch'] = (2.0 * k['FETCH_SIZE'] + k['WRITE_SIZE']) * 1024.0
    json.dump(pmc, open(dst + '_pmc.json', 'w'), indent=1)
    print('\n'.join(lines[:14]))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
