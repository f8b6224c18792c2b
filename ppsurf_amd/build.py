"""Builds libppsurf_amd.so (HIP kernels + C ABI, gfx950 only) in-tree with hipcc.

    python -m ppsurf_amd.build [--force] [--variant NAME -DFLAG ...]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the repo snapshot to the GPU box.
Every source is compiled to its own object (in parallel, only when it or a header changed) and the objects are linked into the
shared library.  `--variant NAME` builds libppsurf_amd_NAME.so with extra -D switches (ablation / tuning builds for tools/;
selected at run time with PPS_LIB_VARIANT=NAME, never used by the product or the tests).
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libppsurf_amd.so')
SOURCES = ['pps_decode.hip', 'pps_knn.hip', 'pps_grow.hip', 'pps_mc.hip', 'pps_mesh.hip', 'pps_fkaconv.hip', 'pps_sample.hip', 'pps_train.hip', 'pps_csr.hip', 'pps_fka_train.hip', 'pps_bn_train.hip',
           'pps_attn_train.hip', 'pps_rows_train.hip', 'pps_gemm_train.hip', 'pps_optim.hip', 'pps_pack.cpp']
HEADERS = ['pps_common.h', 'pps_fka_common.h', 'pps_rows_train_impl.h', 'pps_head_chain_impl.h', os.path.join('..', '..', 'include', 'ppsurf_amd.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC']


def _newer(path, deps):
    if not os.path.isfile(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force=False, verbose=False, variant=None, defines=()):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    lib = LIB if not variant else os.path.join(HERE, 'libppsurf_amd_{}.so'.format(variant))
    objdir = os.path.join(CSRC, 'build' + ('_' + variant if variant else ''))
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    srcs = [s for s in SOURCES if os.path.isfile(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, os.path.splitext(s)[0] + '.o')
        if force or _newer(obj, [src] + headers):
            jobs.append([hipcc] + FLAGS + list(defines) + ['-c', src, '-o', obj])
    if jobs:
        def run(cmd):
            if verbose:
                print(' '.join(cmd), flush=True)
            return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as pool:
            for res in pool.map(run, jobs):
                if res.returncode != 0:
                    sys.stderr.write(res.stdout)
                    raise subprocess.CalledProcessError(res.returncode, res.args)
                if verbose and res.stdout.strip():
                    print(res.stdout)
    objs = [os.path.join(objdir, os.path.splitext(s)[0] + '.o') for s in srcs]
    if jobs or _newer(lib, objs):
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', lib])
    return lib


if __name__ == '__main__':
    argv = sys.argv[1:]
    variant, defines = None, [a for a in argv if a.startswith(('-D', '-f', '-m'))]
    if '--variant' in argv:
        variant = argv[argv.index('--variant') + 1]
    print(build(force='--force' in argv, verbose=True, variant=variant, defines=defines))
