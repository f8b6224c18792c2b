#!/bin/bash
# Round 5, the three bounded inference experiments of VERDICT r4 item 2, each as quick bench lines (JSON) under gpurun_out/r5_infer/:
#   (a) PointNet's per-query matrix M through cache-sized sub-chunks:  PPS_PN_SUB = queries per (stn_fc, feat_rows) pair, PPS_PN_SUB_REUSE = same scratch
#   (b) fc_query with 2 / 1 f16 products instead of 3 (variant builds libppsurf_amd_fcq2.so / _fcq1.so) + the decoder parity tests under them
#   (c) one vs two chunk lanes (PPS_CHUNK_LANES)
# usage (from the repo root, on the GPU box):  bash tools/r5_infer_experiments.sh [steps]
set -u
OUT=gpurun_out/r5_infer
mkdir -p $OUT
STEPS=${1:-94}
run() {   # name, env...
    local name=$1; shift
    env "$@" python bench.py --quick --steps $STEPS --warmup 6 > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name" >> $OUT/failed.txt
    python - "$OUT/$name.json" "$name" <<'EOF'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    sm = d['stage_ms']
    print('{:>22s}: {:6.3f} ms/step {:6.2f} M q/s lanes {} | single {:6.3f} ms | interp {:.3f} stn_rows {:.3f} stn_fc {:.3f} feat {:.3f} tail {:.3f} spatial {:.3f}'.format(
        sys.argv[2], d['ms_per_step'], d['value'] / 1e6, d['lanes'], d['single_lane']['ms_per_step'], sm['interp_pool'], sm['pointnet_stn_rows'],
        sm['pointnet_stn_fc'], sm['pointnet_feat_rows'], sm['decode_tail'], d['spatial_ms']))
except Exception as exc:
    print(sys.argv[2], 'no result:', exc)
EOF
}
run base_auto X=1
run base_lanes1 PPS_CHUNK_LANES=1
run base_lanes2 PPS_CHUNK_LANES=2
run base_lanes3 PPS_CHUNK_LANES=3
for sub in 2048 4096 8192 16384 25000; do
    run sub${sub} PPS_CHUNK_LANES=1 PPS_PN_SUB=$sub
    run sub${sub}_reuse PPS_CHUNK_LANES=1 PPS_PN_SUB=$sub PPS_PN_SUB_REUSE=1
done
run fcq2 PPS_CHUNK_LANES=1 PPS_LIB_VARIANT=fcq2
run fcq1 PPS_CHUNK_LANES=1 PPS_LIB_VARIANT=fcq1
run base_lanes1_again PPS_CHUNK_LANES=1
for v in fcq2 fcq1; do
    PPS_LIB_VARIANT=$v timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_api.py -m gpu -q --no-header -p no:cacheprovider > $OUT/parity_$v.txt 2>&1
    echo "parity $v: $(tail -1 $OUT/parity_$v.txt)"
done
