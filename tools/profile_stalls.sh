#!/bin/bash
# Where do the waves of the decoder kernels spend their cycles?  SQ counter passes (no trace domains) over `bench.py --quick`:
# wave-parked (s_waitcnt / barrier) vs issue-stalled vs issuing, per instruction class; LDS activity and bank conflicts.
#   tools/profile_stalls.sh <tag> [f16x3|f32]   -> gpurun_out/stalls_<tag>/summary_{rocprof_summary.txt,pmc.json}
TAG=${1:-r4}
DT=${2:-f16x3}
OUT=$PWD/gpurun_out/stalls_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 20 --warmup 3 --quick --dtype $DT"
cd /tmp
rocprofv3 --list-avail > $OUT/list_avail.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_wait -o pmc -- $BENCH > $OUT/pmc_wait.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM -d $OUT/pmc_active -o pmc -- $BENCH > $OUT/pmc_active.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES -d $OUT/pmc_insts -o pmc -- $BENCH > $OUT/pmc_insts.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN -d $OUT/pmc_lds -o pmc -- $BENCH > $OUT/pmc_lds.log 2>&1
cd - > /dev/null
python tools/rocpd_summary.py $OUT $OUT/summary > /dev/null
find $OUT -name "*.db" -delete
grep -c . $OUT/summary_rocprof_summary.txt
