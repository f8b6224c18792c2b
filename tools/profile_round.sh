#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, bench, rocprofv3 kernel-trace stats and PMC passes of bench.py.
# Outputs land in gpurun_out/prof_$TAG/ ; copy the summaries you want judged into profiles/.
#   tools/profile_round.sh <tag> [f32|f16x3]
TAG=${1:-r02}
DT=${2:-f32}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python bench.py --quick --dtype $DT > $OUT/bench.json 2> $OUT/bench.err
# [r5] the traced / counted command runs ONE chunk lane: with the product default of two lanes the kernels of two chunks overlap and a kernel's
# traced duration contains its share of the other lane (the bench line's per-kernel times come from its own single-lane pass for the same reason)
BENCH="env PPS_CHUNK_LANES=1 python $PWD/bench.py --steps 20 --warmup 3 --quick --dtype $DT"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o pmc -- $BENCH > $OUT/pmc_mfma.log 2>&1
cd - > /dev/null
# summarise on the box, then drop the rocpd databases (tens of MB) so that the merge back stays small
python tools/rocpd_summary.py $OUT $OUT/summary > /dev/null
find $OUT -name "*.db" -delete
du -sh $OUT
