"""Builds libppsurf_amd.so (HIP kernels + C ABI, gfx950 only) in-tree with hipcc.

    python -m ppsurf_amd.build

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libppsurf_amd.so')
SOURCES = ['pps_decode.hip', 'pps_knn.hip', 'pps_fkaconv.hip', 'pps_sample.hip', 'pps_train.hip', 'pps_fka_train.hip', 'pps_bn_train.hip', 'pps_pack.cpp']
HEADERS = ['pps_common.h', 'pps_fka_common.h', os.path.join('..', '..', 'include', 'ppsurf_amd.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC']


def _stale():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS if os.path.isfile(os.path.join(CSRC, s))]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.isfile(os.path.join(CSRC, s))]
    cmd = [hipcc] + FLAGS + srcs + ['-o', LIB]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
