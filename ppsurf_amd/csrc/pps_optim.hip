// AdamW step of the fit loop over ALL parameter tensors in one launch (gfx950).
//
// replaces: the optimizer of the reference's training run (configs/poco.yaml:60-69 `torch.optim.AdamW`, stepped by the trainer after every
// batch).  torch's fused implementation walks its tensor lists in slabs of 65 536 elements per workgroup: the 13.7 M parameters of PPSurf
// (298 tensors, most of them far smaller than a slab) give ~300 workgroups of work for 256 CUs and nine launches, 0.78 ms per step for
// 384 MB of traffic.  Here the host keeps a device table of 4096-element pieces (one workgroup each, ~3600 of them) with the four
// pointers of every piece; the step is one pass at HBM speed.  Arithmetic = torch's fused AdamW (fp32, amsgrad off, maximize off):
//   g  = grad / grad_scale                      (GradScaler of 16-mixed; written back like torch does)
//   p -= lr * weight_decay * p
//   m  = m + (1 - beta1) (g - m);   v = beta2 v + (1 - beta2) g g
//   p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// with t = the parameter's own step count (a device scalar per parameter as torch keeps it: parameters that got no gradient in some
// step fall behind), advanced by a first tiny kernel unless the scaler found a non-finite gradient (`found_inf`), in which case the
// whole step is skipped.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ppsurf_amd.h"

namespace {

constexpr int OT = 256;

struct Piece {                 // 48 bytes, filled by ppsurf_amd/optim.py
    float* p;
    float* g;
    float* m;
    float* v;
    const float* step;         // the parameter's step count (already advanced for this step)
    int32_t n;
    int32_t pad;
};

__global__ __launch_bounds__(OT) void adamw_tick_kernel(float* const* __restrict__ steps, int n, const float* __restrict__ found_inf) {
    const int i = blockIdx.x * OT + threadIdx.x;
    if (i >= n) return;
    if (found_inf && *found_inf != 0.f) return;
    *steps[i] += 1.f;
}

__global__ __launch_bounds__(OT) void adamw_pieces_kernel(const Piece* __restrict__ pieces, const float* __restrict__ lr_dev, float lr_host,
                                                         float beta1, float beta2, float eps, float weight_decay,
                                                         const float* __restrict__ grad_scale, const float* __restrict__ found_inf) {
    if (found_inf && *found_inf != 0.f) return;
    const Piece pc = pieces[blockIdx.x];
    const float lr = lr_dev ? *lr_dev : lr_host;
    const double t = (double)*pc.step;
    const float bc1 = (float)(1.0 - pow((double)beta1, t));
    const float bc2_sqrt = sqrtf((float)(1.0 - pow((double)beta2, t)));
    const float step_size = lr / bc1;
    const float inv_scale = grad_scale ? 1.f / *grad_scale : 1.f;
    for (int i = threadIdx.x; i < pc.n; i += OT) {
        float g = pc.g[i];
        if (grad_scale) {
            g *= inv_scale;
            pc.g[i] = g;
        }
        float p = pc.p[i], m = pc.m[i], v = pc.v[i];
        p -= lr * weight_decay * p;
        m = m + (1.f - beta1) * (g - m);
        v = beta2 * v + (1.f - beta2) * g * g;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        p -= step_size * m / denom;
        pc.p[i] = p;
        pc.m[i] = m;
        pc.v[i] = v;
    }
}

// 16-bit images of many fp32 tensors in one launch (the autocast copies of all parameters at the start of a training forward pass): the same
// table-of-pieces shape as the optimizer step.  torch's multi-tensor copy takes 11 launches and 0.64 ms for the 298 tensors / 13.7 M parameters of
// PPSurf (most tensors are far smaller than one of its slabs); this is one pass at HBM speed.  Round to nearest even like torch's conversions.
struct CastPiece {             // 24 bytes, filled by ppsurf_amd/train_graph.py
    const float* src;
    uint16_t* dst;
    int32_t n;
    int32_t pad;
};
__device__ __forceinline__ uint16_t to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;                 // NaN (c10::BFloat16)
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint16_t to_f16(float f) {
    const _Float16 h = (_Float16)f;
    return *(const uint16_t*)&h;
}
template <int DT>
__global__ __launch_bounds__(OT) void cast_pieces_kernel(const CastPiece* __restrict__ pieces) {
    const CastPiece pc = pieces[blockIdx.x];
    for (int i = threadIdx.x; i < pc.n; i += OT) pc.dst[i] = DT == 1 ? to_bf16(pc.src[i]) : to_f16(pc.src[i]);
}

}  // namespace

extern "C" {

int pps_cast_piece_bytes(void) { return (int)sizeof(CastPiece); }

int pps_cast_pieces(const void* pieces, int n_pieces, int dtype, void* stream) {
    if (n_pieces < 0 || (dtype != 1 && dtype != 2)) return 1;
    if (n_pieces == 0) return 0;
    if (!pieces) return 1;
    if (dtype == 1) hipLaunchKernelGGL(cast_pieces_kernel<1>, dim3(n_pieces), dim3(OT), 0, (hipStream_t)stream, (const CastPiece*)pieces);
    else hipLaunchKernelGGL(cast_pieces_kernel<2>, dim3(n_pieces), dim3(OT), 0, (hipStream_t)stream, (const CastPiece*)pieces);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int pps_adamw_piece_bytes(void) { return (int)sizeof(Piece); }

int pps_adamw_step(const void* pieces, int n_pieces, const void* steps, int n_steps, const float* lr_dev, float lr, float beta1, float beta2,
                   float eps, float weight_decay, const float* grad_scale, const float* found_inf, void* stream) {
    if (n_pieces < 0 || n_steps < 0) return 1;
    if (n_pieces == 0) return 0;
    if (!pieces || !steps || n_steps == 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adamw_tick_kernel, dim3((n_steps + OT - 1) / OT), dim3(OT), 0, st, (float* const*)steps, n_steps, found_inf);
    hipLaunchKernelGGL(adamw_pieces_kernel, dim3(n_pieces), dim3(OT), 0, st, (const Piece*)pieces, lr_dev, lr, beta1, beta2, eps, weight_decay,
                       grad_scale, found_inf);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"
