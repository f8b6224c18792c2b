"""gpurun_out/stalls_<tag>/summary_pmc.json (tools/profile_stalls.sh) -> profiles/<round>_f16x3_stall_counters.md
    python tools/stall_table.py gpurun_out/stalls_r4f profiles/round4_f16x3_stall_counters.md"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(src + '/summary_pmc.json'))
k = d['kernels']
names = [('interp_pool_f16x3_kernel', 'interp_pool_f16x3'), ('pointnet_stn_rows_kernel<true>', 'stn_rows<f16x3>'),
         ('pointnet_feat_rows_kernel<true>', 'feat_rows<f16x3>'), ('knn_blocked_kernel<1>', 'knn_blocked')]
lines = ['# Where the waves of the decoder kernels spend their cycles (SQ counters, one MI355X, f16x3, `tools/profile_stalls.sh`, sources digest {} at commit {})'.format(
    d['csrc_digest'], d['git_head']), '',
    '`python bench.py --steps 20 --warmup 3 --quick --dtype f16x3` under four separate `rocprofv3 --pmc` passes (no trace domains).  Cycle counters are in',
    'units of 4 clocks; shares are of SQ_WAVE_CYCLES (wave-resident time).  parked = SQ_WAIT_ANY (s_waitcnt / barrier), issue-stalled = SQ_WAIT_INST_ANY',
    '(an instruction is ready but cannot issue: matrix pipe taken, dependency), issuing = SQ_ACTIVE_INST_ANY.  pipe = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM/8 x 1024 SIMDs).', '',
    '| kernel | avg µs | parked | issue-stalled (of which LDS) | issuing | matrix pipe busy | MFMA | other VALU | LDS instr | VMEM rd | LDS array active | bank conflicts |',
    '|---|---|---|---|---|---|---|---|---|---|---|---|']
for n, lab in names:
    v = k[n]
    wc = v['SQ_WAVE_CYCLES']
    pipe = v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024)
    lines.append('| `{}` | {:.0f} | {:.1f} % | {:.1f} % ({:.1f} %) | {:.1f} % | {:.1f} % | {:.1f} M | {:.1f} M | {:.1f} M | {:.2f} M | {:.1f} % of busy cycles | {:.2f} % of them |'.format(
        lab, v['avg_us'], 100 * v['SQ_WAIT_ANY'] / wc, 100 * v['SQ_WAIT_INST_ANY'] / wc, 100 * v['SQ_WAIT_INST_LDS'] / wc, 100 * v['SQ_ACTIVE_INST_ANY'] / wc,
        100 * pipe, v['SQ_INSTS_MFMA'] / 1e6, (v['SQ_INSTS_VALU'] - v['SQ_INSTS_MFMA']) / 1e6, v['SQ_INSTS_LDS'] / 1e6, v['SQ_INSTS_VMEM_RD'] / 1e6,
        100 * v['SQ_LDS_IDX_ACTIVE'] / (v['SQ_BUSY_CYCLES'] * 8), 100 * v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1)))
lines += ['', 'Instruction counts are per launch over all waves (interp: 172.8 M MFMAs = 3456 per query x 50 000).',
          'The blocked kNN at the start of round 4 (0.32 ms): 166.3 M vector instructions per launch, issuing 39.7 %.', '']
open(dst, 'w').write('\n'.join(lines))
print('\n'.join(lines[7:13]))
