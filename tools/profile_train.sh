#!/bin/bash
# rocprofv3 kernel trace of the fit step (tools/time_train_step.py, bf16-mixed, batch 10); summary -> gpurun_out/prof_train_$TAG/
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_train_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/time_train_step.py --bf16 --steps 5"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
cd - > /dev/null
python tools/rocpd_summary.py $OUT $OUT/summary > /dev/null
find $OUT -name "*.db" -delete
grep "ms/step" $OUT/trace.log
head -45 $OUT/summary_rocprof_summary.txt
