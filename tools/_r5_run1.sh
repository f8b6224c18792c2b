timeout 900 python -m pytest tests/test_gpu_decoder.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -5
bash tools/r5_infer_experiments.sh 94 2>&1 | tee gpurun_out/r5_infer_summary.txt
