#!/bin/bash
set -e
cd "$(dirname "$0")/.."
SRC="ppsurf_amd/csrc/pps_decode.hip ppsurf_amd/csrc/pps_knn.hip ppsurf_amd/csrc/pps_fkaconv.hip ppsurf_amd/csrc/pps_sample.hip ppsurf_amd/csrc/pps_train.hip ppsurf_amd/csrc/pps_fka_train.hip ppsurf_amd/csrc/pps_bn_train.hip ppsurf_amd/csrc/pps_pack.cpp"
cp ppsurf_amd/libppsurf_amd.so /tmp/lib_orig.so
for V in "-DNT=256" "-DNT=512" "-DNT=512 -DPPS_PRIO=0"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -shared -fPIC $V $SRC -o ppsurf_amd/libppsurf_amd.so
  echo "== $V"; python tools/time_kernels.py 2>&1 | grep "^interp_pool  \|^pointnet_stn_rows  \|^pointnet_feat_rows  "
done
cp /tmp/lib_orig.so ppsurf_amd/libppsurf_amd.so
