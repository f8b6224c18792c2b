#!/bin/bash
# kernel trace of the latent loop (100 encoder passes of a 100k-point cloud, batches of 25 subsets): which kernels the 57 ms per shape are
#   tools/profile_latent_trace.sh <tag>  -> gpurun_out/latent_<tag>/summary_rocprof_summary.txt
TAG=${1:-r4}
OUT=$PWD/gpurun_out/latent_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/time_latent_loop.py"
$CMD > $OUT/plain.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
cd - > /dev/null
python tools/rocpd_summary.py $OUT $OUT/summary > /dev/null
find $OUT -name "*.db" -delete
cat $OUT/plain.log | tail -8
