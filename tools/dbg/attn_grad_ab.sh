export TMPDIR=/tmp
R=$PWD
for mode in rebuilt stored; do
  O=$R/gpurun_out/attn_$mode; mkdir -p $O
  (cd /tmp && PPS_ATTN_GRAD=$mode rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/tools/time_fit_graph.py --steps 10 > $O/trace.log 2>&1)
  python tools/rocpd_summary.py $O $O/summary > /dev/null
  find $O -name "*.db" -delete
  echo "== $mode"; grep "ms/step" $O/trace.log
  grep -E "attn_pool_bwd|rows_layer_kernel<64, 256, true|rows_layer_kernelILi64ELi256ELb1" $O/summary* | head
done
python -m pytest tests/test_gpu_head_chain.py -x -q -m gpu -k rebuilt 2>&1 | tail -2
