"""rocprofv3 --pmc passes of the fit step (tools/profile_train_pmc.sh) -> per-step totals for bench.py's fit.roofline:
executed matrix flops (SQ_INSTS_VALU_MFMA_MOPS_* x 512), HBM bytes (2 x FETCH_SIZE + WRITE_SIZE KiB: gfx950 reports half of a wide coalesced read
stream, MI355X_MICROARCH.md section HBM), matrix-pipe busy share, and the kernels that move the most bytes.

    python tools/train_pmc_summary.py gpurun_out/prof_train_<tag> <steps traced> out.json
"""
import glob
import json
import os
import sqlite3
import subprocess
import sys


def short(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]


def totals(src, sub):
    dbs = glob.glob(os.path.join(src, sub, '*.db'))
    if not dbs:
        return {}, {}
    c = sqlite3.connect(dbs[0])
    tot, per = {}, {}
    for name, counter, n, s in c.execute('select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name'):
        tot[counter] = tot.get(counter, 0.0) + s
        per.setdefault(short(name), {})[counter] = (n, s)
    return tot, per


def main(src, steps, dst):
    steps = int(steps)
    tot, per = {}, {}
    for sub in ('pmc_fetch', 'pmc_write', 'pmc_mfma', 'pmc_mfmabf16'):
        t, p = totals(src, sub)
        tot.update(t)
        for k, v in p.items():
            per.setdefault(k, {}).update(v)
    mops = sum(tot.get('SQ_INSTS_VALU_MFMA_MOPS_' + t, 0.0) for t in ('F32', 'F16', 'BF16'))
    hbm = (2.0 * tot.get('FETCH_SIZE', 0.0) + tot.get('WRITE_SIZE', 0.0)) * 1024.0
    busy = tot.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    gui = tot.get('GRBM_GUI_ACTIVE', 0.0)
    kern = []
    for k, v in per.items():
        b = (2.0 * v.get('FETCH_SIZE', (0, 0.0))[1] + v.get('WRITE_SIZE', (0, 0.0))[1]) * 1024.0 / steps
        f = sum(v.get('SQ_INSTS_VALU_MFMA_MOPS_' + t, (0, 0.0))[1] for t in ('F32', 'F16', 'BF16')) * 512.0 / steps
        kern.append({'kernel': k, 'launches_per_step': v.get('FETCH_SIZE', v.get('WRITE_SIZE', (0, 0)))[0] / steps, 'hbm_bytes_per_step': b, 'mfma_flops_per_step': f})
    kern.sort(key=lambda r: -r['hbm_bytes_per_step'])
    head = os.environ.get('PPS_GIT_HEAD')                 # the GPU box has no .git: the caller passes the commit of the snapshot
    if not head:
        try:
            head = subprocess.check_output(['git', 'rev-parse', '--short=12', 'HEAD'], cwd=os.path.dirname(os.path.abspath(__file__)), text=True,
                                           stderr=subprocess.DEVNULL).strip()
        except Exception:
            head = 'unrecorded'
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    import bench_workloads
    out = {'source': os.path.basename(src.rstrip('/')), 'git_head': head, 'steps_traced': steps, 'fit_digest': bench_workloads.fit_digest(),
           'command': 'python tools/time_fit_graph.py ({} calls of the bench leg\'s FitStep: HIP-graph replay + loader thread; B=10 x 10000 points x 2000 queries, P=50, bf16-mixed)'.format(steps),
           'mfma_flops_per_step': mops * 512.0 / steps, 'mfma_mops_by_type_per_step': {t: tot.get('SQ_INSTS_VALU_MFMA_MOPS_' + t, 0.0) / steps for t in ('F32', 'F16', 'BF16')},
           'hbm_bytes_per_step': hbm / steps, 'fetch_kib_per_step': tot.get('FETCH_SIZE', 0.0) / steps, 'write_kib_per_step': tot.get('WRITE_SIZE', 0.0) / steps,
           'mfma_busy_share_of_kernel_time': busy / (gui / 8.0 * 1024.0) if gui else None,
           'bound': 'hbm', 'top_kernels_by_hbm_bytes': kern[:25]}
    json.dump(out, open(dst, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:4])
