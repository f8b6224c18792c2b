#!/bin/bash
# rocprofv3 passes over the fit step for the fit.roofline object of the bench line: kernel trace + separate --pmc passes
# (FETCH_SIZE | WRITE_SIZE | MFMA op counters | matrix-pipe busy cycles) of `python tools/time_fit_graph.py` (the step replayed as a HIP graph with the
# loader thread on its side stream, exactly the bench leg).  Never --pmc together with a trace domain.
#   tools/profile_train_pmc.sh <tag>      -> gpurun_out/prof_train_<tag>/{summary_rocprof_summary.txt, train_pmc.json}; copy into profiles/
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_train_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# [r4] the REPLAYED step (what bench.py's fit_ms_per_step leg and `pps.py fit` run): 8 warm-up calls (3 eager, the capture, replays) + 3 x STEPS replays
STEPS=10
CMD="python $PWD/tools/time_fit_graph.py --steps $STEPS"
TRACED=$((8 + 3 * STEPS))
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o pmc -- $CMD > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/pmc_mfmabf16 -o pmc -- $CMD > $OUT/pmc_mfmabf16.log 2>&1
cd - > /dev/null
python tools/rocpd_summary.py $OUT $OUT/summary > /dev/null
python tools/train_pmc_summary.py $OUT $TRACED $OUT/train_pmc.json
find $OUT -name "*.db" -delete
grep "ms/step" $OUT/trace.log
cat $OUT/train_pmc.json | head -40
