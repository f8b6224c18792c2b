// Region-growing driver of the iso-surface extraction on byte masks (source/poco_utils.py:178-254): box dilation and the frontier test.
//
// The reference dilates with a Python loop over points (`_dilate_binary`, :181-196: every point marks the box [p - r, p + r], clipped at the
// volume border); round 2 used max_pool3d over a float copy of the mask (1.1 ms per call at R = 257, 12 calls per shape: three quarters of the
// driver's growth time).  Here the masks stay bytes: a separable OR over 2r+1 neighbours along z, y, x -- three streaming passes over the
// (R+2)^3 bytes, HBM/L2-bound (17 MB per pass at R = 257).
#include "pps_common.h"
#include "../../include/ppsurf_amd.h"

namespace {

// dst[i] = OR_{|d| <= r, 0 <= c + d < extent} src[i + d * stride]   with c = (i / stride) % extent; 4 voxels along z per thread
__global__ __launch_bounds__(256) void dilate_axis_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int64_t total, int64_t stride,
                                                          int extent, int r) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= total) return;
    if (stride == 1) {
        // the contiguous axis: row and column are taken per voxel (extent need not be a multiple of 4, and may be smaller than 4: the four
        // voxels of a thread can then span more than two rows)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = i0 + j;
            if (i >= total) break;
            const int c = (int)(i % extent);
            const int64_t base = i - c;
            const int lo = c - r < 0 ? 0 : c - r, hi = c + r >= extent ? extent - 1 : c + r;
            uint8_t v = 0;
            for (int k = lo; k <= hi; ++k) v |= src[base + k];
            dst[i] = v;
        }
        return;
    }
    if (i0 + 3 < total && (stride & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 3) == 0) {
        // strided axis, stride a multiple of 4, both masks 4-byte aligned (a bool tensor with a storage offset need not be): the four voxels are
        // neighbours along z with the same coordinate on this axis -> 32-bit accesses
        const int c = (int)((i0 / stride) % extent);
        const int lo = c - r < 0 ? -c : -r, hi = c + r >= extent ? extent - 1 - c : r;
        unsigned v = 0;
        for (int d = lo; d <= hi; ++d) v |= *(const unsigned*)(src + i0 + (int64_t)d * stride);
        *(unsigned*)(dst + i0) = v;
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = i0 + j;
        if (i >= total) break;
        const int c = (int)((i / stride) % extent);
        const int lo = c - r < 0 ? -c : -r, hi = c + r >= extent ? extent - 1 - c : r;
        uint8_t v = 0;
        for (int d = lo; d <= hi; ++d) v |= src[i + (int64_t)d * stride];
        dst[i] = v;
    }
}

// The frontier of one growth round (poco_utils.py:240-246), after the seeds' dilations:
//   new[i] = to_see[i] && ((neg[i] && vol[i] >= 0) || (pos[i] && vol[i] <= 0))        (NaN = not evaluated: both comparisons false)
__global__ __launch_bounds__(256) void frontier_kernel(const double* __restrict__ vol, const uint8_t* __restrict__ neg, const uint8_t* __restrict__ pos,
                                                       const uint8_t* __restrict__ to_see, uint8_t* __restrict__ out, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const double v = vol[i];
    out[i] = (to_see[i] && ((neg[i] && v >= 0.0) || (pos[i] && v <= 0.0))) ? 1 : 0;
}

// todo[i] = band[i] && isnan(vol[i])   (voxels of the dilated band that have not been evaluated yet)
__global__ __launch_bounds__(256) void band_todo_kernel(const double* __restrict__ vol, const uint8_t* __restrict__ band, uint8_t* __restrict__ out, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    out[i] = (band[i] && vol[i] != vol[i]) ? 1 : 0;
}

}  // namespace

extern "C" {

int pps_dilate_box_u8(const uint8_t* src, uint8_t* dst, uint8_t* tmp, int64_t nx, int64_t ny, int64_t nz, int r, void* stream) {
    if (nx < 1 || ny < 1 || nz < 1 || r < 0) return PPS_ERR_ARG;
    if (!src || !dst || !tmp || src == dst || src == tmp || dst == tmp) return PPS_ERR_ARG;
    const int64_t total = nx * ny * nz;
    const unsigned blocks = (unsigned)((total + 1023) / 1024);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dilate_axis_kernel, dim3(blocks), dim3(256), 0, st, src, dst, total, (int64_t)1, (int)nz, r);
    hipLaunchKernelGGL(dilate_axis_kernel, dim3(blocks), dim3(256), 0, st, (const uint8_t*)dst, tmp, total, nz, (int)ny, r);
    hipLaunchKernelGGL(dilate_axis_kernel, dim3(blocks), dim3(256), 0, st, (const uint8_t*)tmp, dst, total, ny * nz, (int)nx, r);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_grow_frontier_f64(const double* vol, const uint8_t* neg, const uint8_t* pos, const uint8_t* to_see, uint8_t* out, int64_t total, void* stream) {
    if (total < 0) return PPS_ERR_ARG;
    if (total == 0) return PPS_OK;
    if (!vol || !neg || !pos || !to_see || !out) return PPS_ERR_ARG;
    hipLaunchKernelGGL(frontier_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vol, neg, pos, to_see, out, total);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_grow_band_todo_f64(const double* vol, const uint8_t* band, uint8_t* out, int64_t total, void* stream) {
    if (total < 0) return PPS_ERR_ARG;
    if (total == 0) return PPS_OK;
    if (!vol || !band || !out) return PPS_ERR_ARG;
    hipLaunchKernelGGL(band_todo_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vol, band, out, total);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

}  // extern "C"
