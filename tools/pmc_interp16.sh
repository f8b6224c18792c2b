#!/bin/bash
# What bounds interp_pool_f16x3_kernel?  LDS / wait counters of the f16x3 bench in separate --pmc passes (never with a trace domain).
#   tools/pmc_interp16.sh <tag>  -> gpurun_out/pmc_interp16_<tag>/summary.txt
TAG=${1:-r03}
OUT=$PWD/gpurun_out/pmc_interp16_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 10 --warmup 2 --quick --dtype f16x3"
cd /tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_MEM_VIOLATIONS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_FLAT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/pmc_$i -o pmc -- $BENCH > $OUT/pmc_$i.log 2>&1
done
cd - > /dev/null
python - "$OUT" <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
lines = []
for sub in sorted(glob.glob(os.path.join(out, 'pmc_*'))):
    dbs = glob.glob(os.path.join(sub, '*.db'))
    if not os.path.isdir(sub) or not dbs:
        continue
    c = sqlite3.connect(dbs[0])
    for name, counter, n, avg in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if 'interp_pool' in name or 'feat_rows' in name or 'stn_rows' in name:
            lines.append('{:44s} {:34s} n={:3d} avg={:.6g}'.format(name.split('(')[0][-44:], counter, n, avg))
open(os.path.join(out, 'summary.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
find $OUT -name "*.db" -delete
grep -c . $OUT/avail.txt
