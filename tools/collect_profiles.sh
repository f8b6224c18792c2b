#!/bin/bash
# Copies the summaries of tools/profile_round.sh / profile_train_pmc.sh runs (gpurun_out/prof_<tag>_{f16x3,f32}, gpurun_out/prof_train_<tag>) into
# profiles/<round>_* -- the files bench.py reads for roofline.traffic and fit.roofline (digest-checked) and the ones DESIGN.md cites.
#   tools/collect_profiles.sh r4c round4
TAG=$1; RND=$2
for dt in f16x3 f32; do
  src=gpurun_out/prof_${TAG}_$dt
  [ -d $src ] || continue
  cp $src/summary_rocprof_summary.txt profiles/${RND}_${dt}_rocprof_summary.txt
  cp $src/summary_pmc.json profiles/${RND}_${dt}_pmc.json
  cp $src/bench.json profiles/${RND}_${dt}_bench_quick.json
  python tools/kernel_table.py profiles/${RND}_${dt} > /dev/null
done
src=gpurun_out/prof_train_$TAG
if [ -d $src ]; then
  cp $src/train_pmc.json profiles/${RND}_train_pmc.json
  cp $src/summary_rocprof_summary.txt profiles/${RND}_train_rocprof_summary.txt
fi
