"""profiles/<prefix>_pmc.json -> profiles/<prefix>_kernel_table.md (per-kernel HBM traffic, MFMA utilisation, clock)."""
import json
import sys

ORDER = ['interp_pool_kernel', 'interp_pool_f16x3_kernel', 'pointnet_feat_rows_kernel<false>', 'pointnet_feat_rows_kernel<true>',
         'pointnet_stn_rows_kernel<false>', 'pointnet_stn_rows_kernel<true>', 'pointnet_stn_fc_kernel', 'pointnet_stn_fc_h_kernel',
         'pointnet_feat_rows_kernel', 'pointnet_stn_rows_kernel', 'knn_blocked_kernel<1>', 'decode_tail_kernel', 'decode_tail_h_kernel', 'patch_normalize_kernel',
         'rows_dense256_kernel']


def main(prefix):
    k = json.load(open(prefix + '_pmc.json'))['kernels']
    # a kernel whose name the profiler did not demangle (half8 arguments) is listed under its mangled name
    k = dict(k, **{'pointnet_stn_fc_h_kernel': v for n, v in k.items() if 'pointnet_stn_fc_h_kernel' in n})
    import os
    tag = os.path.basename(prefix)
    steps = None
    for nm in ('interp_pool_f16x3_kernel', 'interp_pool_kernel'):
        if nm in k and 'calls' in k[nm]:
            steps = k[nm]['calls']                       # one launch of the interpolation kernel per bench step
            break
    lines = ['# Per-kernel time, HBM traffic and MFMA utilisation ({}, one MI355X)'.format(tag), '',
             'Source: `{0}_rocprof_summary.txt` / `{0}_pmc.json` (`tools/profile_round.sh`: `rocprofv3 --kernel-trace --stats` and separate'.format(tag),
             '`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc SQ_*/GRBM_GUI_ACTIVE` passes of `python bench.py --steps 20 --warmup 3 --quick`).',
             'FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read stream, MI355X_MICROARCH.md §HBM); KiB -> MB.',
             'MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs).',
             '**Per-STEP sums**: a bench step launches some kernels more than once (the PointNet row kernels: packed full rounds + per-query remainder);',
             '`launches/step` = dispatches of the kernel / dispatches of the interpolation kernel ({} steps traced), and the per-step columns are'.format(steps),
             'per-dispatch average x launches/step.', '',
             '| kernel | launches/step | µs per step | HBM read MB per step | HBM write MB per step | HBM TB/s | MFMA util | clock GHz |', '|---|---|---|---|---|---|---|---|']
    tot = [0.0, 0.0, 0.0]
    for name in ORDER:
        v = k.get(name)
        if not v or 'FETCH_SIZE' not in v or 'avg_us' not in v:       # (a kernel below the trace summary's 0.005 % cut has counters but no time)
            continue
        us = v['avg_us'] / 1e3 if v['avg_us'] > 1e5 else v['avg_us']
        per = (v.get('calls', steps) / steps) if steps else 1.0
        rd, wr = 2 * v['FETCH_SIZE'] * 1024 / 1e6, v['WRITE_SIZE'] * 1024 / 1e6
        cyc = v['GRBM_GUI_ACTIVE'] / 8.0
        util = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cyc * 1024.0)
        lines.append('| `{}` | {:.2f} | {:.1f} | {:.1f} | {:.1f} | {:.3f} | {:.1f} % | {:.2f} |'.format(name, per, us * per, rd * per, wr * per, (rd + wr) / us, 100 * util,
                                                                                                   cyc / (us * 1e3)))
        tot[0] += us * per; tot[1] += rd * per; tot[2] += wr * per
    lines.append('| **sum of the listed kernels** | | {:.1f} | {:.1f} | {:.1f} | | | |'.format(*tot))
    lines += ['', 'MFMA utilisation counts BUSY cycles of the matrix pipe whatever the operand type (fp32 MFMA in the f32 run, f16 MFMA in the',
              'f16x3 run).  `knn_blocked_kernel` is VALU/selection-bound with its 1.2 MB cloud in L2.  FETCH_SIZE / WRITE_SIZE count L2 <-> fabric',
              'requests whether the Infinity Cache or HBM serves them (MI355X_MICROARCH.md).']
    open(prefix + '_kernel_table.md', 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[9:20]))


if __name__ == '__main__':
    main(sys.argv[1])
