#!/bin/bash
# Counter passes of the config-5 leg (BASELINE config 5: N = 250k, P = 200, Q = 25 000): rocprofv3 kernel trace + separate
# --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py --only config5` with ONE chunk lane -> gpurun_out/prof_c5_<tag>/config5_pmc.json
# (HBM bytes per launch of the k = 200 block search and of the patch kernel + the digest of the kernel sources); copy it to
# profiles/round6_config5_pmc.json.  Never --pmc together with a trace domain.
#   tools/profile_config5.sh <tag>
TAG=${1:-r06}
OUT=$PWD/gpurun_out/prof_c5_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="env PPS_CHUNK_LANES=1 python $PWD/bench.py --only config5"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
cd - > /dev/null
python tools/rocpd_summary.py $OUT $OUT/summary > /dev/null
python tools/config5_pmc_summary.py $OUT $OUT/config5_pmc.json
find $OUT -name "*.db" -delete
cat $OUT/config5_pmc.json
