#!/bin/bash
# builds ablated variants of the decoder library and times interp_pool with each (development aid)
set -e
cd "$(dirname "$0")/.."
SRC="ppsurf_amd/csrc/pps_decode.hip ppsurf_amd/csrc/pps_knn.hip ppsurf_amd/csrc/pps_fkaconv.hip ppsurf_amd/csrc/pps_pack.cpp"
cp ppsurf_amd/libppsurf_amd.so /tmp/lib_orig.so
for V in BASE NOSTREAM NOBARRIER NOGATHER NOSOFTMAX "NOSTREAM -DPPS_ABL_NOBARRIER -DPPS_ABL_NOGATHER -DPPS_ABL_NOSOFTMAX"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -shared -fPIC -DPPS_ABL_$V $SRC -o ppsurf_amd/libppsurf_amd.so
  echo "== $V"; python tools/time_kernels.py 2>&1 | grep "^interp_pool  "
done
cp /tmp/lib_orig.so ppsurf_amd/libppsurf_amd.so
