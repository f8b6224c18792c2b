# kernel trace of the fit step with the PointNet attention gradient rebuilt on load (default) and stored (PPS_PATCH_ATTN_GRAD=stored)
export TMPDIR=/tmp
R=$PWD
for mode in rebuilt stored; do
  O=$R/gpurun_out/pattn_$mode; mkdir -p $O
  (cd /tmp && PPS_PATCH_ATTN_GRAD=$mode rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/tools/time_fit_graph.py --steps 10 > $O/trace.log 2>&1)
  python tools/rocpd_summary.py $O $O/summary > /dev/null
  find $O -name "*.db" -delete
  echo "== $mode"; grep "ms/step" $O/trace.log
  grep -E "patch_attn_bwd|rows_layer_kernel<256, 128, true|rows_dw_kernel<128, 256, true" $O/summary_rocprof_summary.txt | head
done
