// Neighbourhood gather / scatter kernels of the TRAINING step, forward and backward (gfx950).
//
// The dense layers of a fit step are library GEMMs over all rows of the batch; what is specific to this network is the
// neighbourhood traffic: gather x rows through an id table, contract with the 16x16 kernel weights of the FKAConv layer,
// max-pool over a neighbourhood, gather latent rows for the interpolation head -- and the transposes of all of these in the
// backward pass, which are scatter-adds.  Scatter-adds are done WITHOUT atomics: the caller sorts the id table once per step
// (stable sort -> `order`, `offsets`: CSR of "which (m,j) entries point at row n") and every output row sums its own
// contributions in a fixed order, so gradients are bit-reproducible from run to run.
//
// All kernels are HBM/L2-bandwidth bound streaming kernels: one thread per (row, channel) with the channel fastest, so every
// row access is a contiguous C*4-byte segment; id tables are read as wave-wide broadcasts.
//
// replaces (reference, under autograd): source/base/nn.py:655-674 `batch_gather` (+ its index_add backward), :677-680
// `max_pool`, :647-649 the FKAConv feature aggregation, source/poco_model.py:400 the latent gather of the interpolation head.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ppsurf_amd.h"

namespace {

constexpr int TB = 256;
constexpr int KT = 16;          // FKAConv kernel size (columns of the weighting matrix)

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 2; }
inline unsigned blocks_for(int64_t threads) { return (unsigned)((threads + TB - 1) / TB); }

__device__ __forceinline__ void vadd(float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
__device__ __forceinline__ void vadd(float& a, const float& v) { a += v; }
__device__ __forceinline__ void vzero(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void vzero(float& a) { a = 0.f; }

// out[r, :] = x[idx[r], :]           (V = float4 when C % 4 == 0, else float; c4 = row length in units of V)
template <typename V>
__global__ void __launch_bounds__(TB) gather_rows_kernel(const V* __restrict__ x, const int64_t* __restrict__ idx,
                                                         int64_t r, int c4, V* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= r * c4) return;
    const int64_t row = t / c4;
    const int col = (int)(t - row * c4);
    out[t] = x[idx[row] * c4 + col];
}

// out[n, :] = sum over e in [offsets[n], offsets[n+1]) of vals[order[e], :]   (fixed order: deterministic)
template <typename V>
__global__ void __launch_bounds__(TB) segment_sum_rows_kernel(const V* __restrict__ vals, const int64_t* __restrict__ order,
                                                              const int64_t* __restrict__ offsets, int64_t n, int c4,
                                                              V* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n * c4) return;
    const int64_t row = t / c4;
    const int col = (int)(t - row * c4);
    V acc;
    vzero(acc);
    const int64_t e1 = offsets[row + 1];
    for (int64_t e = offsets[row]; e < e1; ++e) vadd(acc, vals[order[e] * c4 + col]);
    out[t] = acc;
}

// the same with 16-bit values (bfloat16, or IEEE half with F16; 4 channels = 8 bytes per thread), fp32 accumulation and output
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
template <bool F16>
__global__ void __launch_bounds__(TB) segment_sum_rows_16_kernel(const uint2* __restrict__ vals, const int64_t* __restrict__ order,
                                                                 const int64_t* __restrict__ offsets, int64_t n, int c4,
                                                                 float4* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n * c4) return;
    const int64_t row = t / c4;
    const int col = (int)(t - row * c4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t e1 = offsets[row + 1];
    for (int64_t e = offsets[row]; e < e1; ++e) {
        const uint2 v = vals[order[e] * c4 + col];
        if (F16) {
            const half2v a = *(const half2v*)&v.x, b = *(const half2v*)&v.y;
            acc.x += (float)a[0]; acc.y += (float)a[1]; acc.z += (float)b[0]; acc.w += (float)b[1];
        } else {
            acc.x += __uint_as_float(v.x << 16); acc.y += __uint_as_float(v.x & 0xffff0000u);
            acc.z += __uint_as_float(v.y << 16); acc.w += __uint_as_float(v.y & 0xffff0000u);
        }
    }
    out[t] = acc;
}

// FKAConv feature aggregation: out[m, c*16 + t] = sum_j x[idx[m,j], c] * g[m,j,t]
__global__ void __launch_bounds__(TB) contract_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                          const float* __restrict__ g, int64_t m, int k, int c,
                                                          float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= m * c) return;
    const int64_t row = t / c;
    const int ch = (int)(t - row * c);
    float acc[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i) acc[i] = 0.f;
    const float4* gm = (const float4*)(g + row * k * KT);
    for (int j = 0; j < k; ++j) {
        const float xv = x[idx[row * k + j] * c + ch];
#pragma unroll
        for (int i = 0; i < KT / 4; ++i) {
            const float4 w = gm[j * (KT / 4) + i];
            acc[4 * i + 0] += xv * w.x; acc[4 * i + 1] += xv * w.y; acc[4 * i + 2] += xv * w.z; acc[4 * i + 3] += xv * w.w;
        }
    }
    float4* o = (float4*)(out + t * KT);
#pragma unroll
    for (int i = 0; i < KT / 4; ++i) o[i] = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
}

// backward, part 1: per gathered entry  dxg[m,j,c] = sum_t dout[m, c*16+t] * g[m,j,t]   (scattered to x rows by segment_sum)
__global__ void __launch_bounds__(TB) contract_bwd_x_kernel(const float* __restrict__ dout, const float* __restrict__ g,
                                                            int64_t m, int k, int c, float* __restrict__ dxg) {
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= m * c) return;
    const int64_t row = t / c;
    const int ch = (int)(t - row * c);
    float d[KT];
    const float4* dp = (const float4*)(dout + t * KT);
#pragma unroll
    for (int i = 0; i < KT / 4; ++i) { const float4 v = dp[i]; d[4 * i] = v.x; d[4 * i + 1] = v.y; d[4 * i + 2] = v.z; d[4 * i + 3] = v.w; }
    const float4* gm = (const float4*)(g + row * k * KT);
    for (int j = 0; j < k; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < KT / 4; ++i) {
            const float4 w = gm[j * (KT / 4) + i];
            s += d[4 * i] * w.x + d[4 * i + 1] * w.y + d[4 * i + 2] * w.z + d[4 * i + 3] * w.w;
        }
        dxg[(row * k + j) * c + ch] = s;
    }
}

// backward, part 2: dg[m,j,t] = sum_c dout[m, c*16+t] * x[idx[m,j], c]      one thread per (m, j, t), t fastest
__global__ void __launch_bounds__(TB) contract_bwd_g_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                            const int64_t* __restrict__ idx, int64_t m, int k, int c,
                                                            float* __restrict__ dg) {
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= m * k * KT) return;
    const int tt = (int)(t % KT);
    const int64_t mj = t / KT;
    const int64_t row = mj / k;
    const float* xr = x + idx[mj] * c;
    const float* dr = dout + row * c * KT + tt;
    float s = 0.f;
    for (int ch = 0; ch < c; ++ch) s += dr[ch * KT] * xr[ch];
    dg[t] = s;
}

// ---- the same three products on the matrix pipe (c a multiple of 16): per support point they are 16 x c x 16 matrix products, and the
// operands can be loaded straight from global memory in v_mfma_f32_16x16x4_f32 fragment order (the contraction index may be permuted
// freely, so a lane takes FOUR consecutive values with one 16-byte load and feeds them to four MFMA steps).  One wave per point.
// fp32 MFMA is an exact fmaf chain: same arithmetic as the thread-per-output kernels above, which remain for c = 3 (the first layer).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int CW = 4;           // waves (= support points) per workgroup iteration

// Storage of x, out and dout: float, bfloat16 (uint16_t) or IEEE half (_Float16) -- the 16-bit activations of an autocast step are read and
// written as they are (no cast kernels around the op, half the traffic); the products and g / dg / dxg are fp32.
typedef _Float16 half_c;
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const uint16_t* p) { return __uint_as_float((uint32_t)*p << 16); }
__device__ __forceinline__ float ld1(const half_c* p) { return (float)*p; }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 ld4(const uint16_t* p) {
    const uint2 t = *(const uint2*)p;
    return f32x4{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 ld4(const half_c* p) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 t = *(const h4*)p;
    return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
}
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(uint16_t* p, float v) {
    uint32_t u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);                     // round to nearest even (finite values)
    *p = (uint16_t)(u >> 16);
}
__device__ __forceinline__ void st1(half_c* p, float v) { *p = (half_c)v; }

// out[m, c*16 + t] = sum_j x[idx[m,j], c] g[m,j,t]:   D[c'][t] over 16-channel blocks, contraction over the neighbours j = 4 s + q
template <typename T>
__global__ void __launch_bounds__(CW * 64) contract_fwd_mfma_kernel(const T* __restrict__ x, const int64_t* __restrict__ idx,
                                                                    const float* __restrict__ g, int64_t m, int k, int c, T* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    for (int64_t row = (int64_t)blockIdx.x * CW + wave; row < m; row += (int64_t)gridDim.x * CW) {
        float gb[4];
        int64_t xr[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int j = 4 * s + q;
            gb[s] = j < k ? g[(row * k + j) * KT + n] : 0.f;                     // B[j][t = n]
            xr[s] = j < k ? idx[row * k + j] * c : -1;
        }
        for (int cb = 0; cb < c; cb += 16) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float a = xr[s] >= 0 ? ld1(x + xr[s] + cb + n) : 0.f;      // A[c' = n][j]
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, gb[s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) st1(out + (row * c + cb + 4 * q + r) * KT + n, acc[r]);      // D row 4 q + r = channel, column n = t
        }
    }
}

// dg[m,j,t] = sum_c dout[m, c*16+t] x[idx[m,j], c]:   D[j][t], contraction over the channels c = 16 cb + 4 q + s
template <typename T>
__global__ void __launch_bounds__(CW * 64) contract_bwd_g_mfma_kernel(const T* __restrict__ dout, const T* __restrict__ x,
                                                                      const int64_t* __restrict__ idx, int64_t m, int k, int c,
                                                                      float* __restrict__ dg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    for (int64_t row = (int64_t)blockIdx.x * CW + wave; row < m; row += (int64_t)gridDim.x * CW) {
        const int64_t xr = n < k ? idx[row * k + n] * c : -1;                    // A row j = n
        const T* dr = dout + row * c * KT + n;                                   // B column t = n
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int cb = 0; cb < c; cb += 16) {
            const f32x4 a4 = xr >= 0 ? ld4(x + xr + cb + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s], ld1(dr + (cb + 4 * q + s) * KT), acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < k) dg[(row * k + 4 * q + r) * KT + n] = acc[r];
    }
}

// dxg[m,j,c] = sum_t dout[m, c*16+t] g[m,j,t]:   D[c'][j], contraction over t = 4 q + s
template <typename T>
__global__ void __launch_bounds__(CW * 64) contract_bwd_x_mfma_kernel(const T* __restrict__ dout, const float* __restrict__ g, int64_t m, int k,
                                                                      int c, float* __restrict__ dxg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    for (int64_t row = (int64_t)blockIdx.x * CW + wave; row < m; row += (int64_t)gridDim.x * CW) {
        const f32x4 b4 = n < k ? *(const f32x4*)(g + (row * k + n) * KT + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};      // B[t][j = n]
        for (int cb = 0; cb < c; cb += 16) {
            const f32x4 a4 = ld4(dout + (row * c + cb + n) * KT + 4 * q);                                           // A[c' = n][t]
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s], b4[s], acc, 0, 0, 0);
            if (n < k) *(f32x4*)(dxg + (row * k + n) * c + cb + 4 * q) = acc;    // D rows 4 q + r = channels cb + 4 q + r, column n = j
        }
    }
}

inline unsigned point_blocks(int64_t m) {
    const int64_t b = (m + CW - 1) / CW;
    return (unsigned)(b < 8192 ? b : 8192);
}

template <typename T>
int contract_fwd_t(const T* x, const int64_t* idx, const float* g, int64_t m, int k, int c, T* out, hipStream_t st) {
    contract_fwd_mfma_kernel<T><<<point_blocks(m), CW * 64, 0, st>>>(x, idx, g, m, k, c, out);
    return launch_status();
}
template <typename T>
int contract_bwd_t(const T* x, const int64_t* idx, const float* g, const T* dout, int64_t m, int k, int c, float* dxg, float* dg,
                          hipStream_t st) {
    if (dxg) contract_bwd_x_mfma_kernel<T><<<point_blocks(m), CW * 64, 0, st>>>(dout, g, m, k, c, dxg);
    if (dg) contract_bwd_g_mfma_kernel<T><<<point_blocks(m), CW * 64, 0, st>>>(dout, x, idx, m, k, c, dg);
    return launch_status();
}


// out[m,c] = max_j x[idx[m,j], c], arg[m,c] = first j attaining it.  T: float, or a 16-bit storage type (the maximum of 16-bit values is one of
// them: the result is exact in the type it came in)
template <typename T>
__global__ void __launch_bounds__(TB) gather_max_arg_kernel(const T* __restrict__ x, const int64_t* __restrict__ idx,
                                                            int64_t m, int k, int c, T* __restrict__ out,
                                                            int32_t* __restrict__ arg) {
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= m * c) return;
    const int64_t row = t / c;
    const int ch = (int)(t - row * c);
    int64_t at = idx[row * k] * c + ch;
    float best = ld1(x + at);
    int bj = 0;
    for (int j = 1; j < k; ++j) {
        const int64_t a = idx[row * k + j] * c + ch;
        const float v = ld1(x + a);
        if (v > best) { best = v; bj = j; at = a; }
    }
    out[t] = x[at];
    arg[t] = bj;
}

// dx[n,c] = sum over entries e=(m,j) pointing at n (CSR order) of dout[m,c] * [arg[m,c] == j]
//   (T a 16-bit type: gradients read in it, summed in fp32 in CSR order, the sum rounded to it once)
template <typename T>
__global__ void __launch_bounds__(TB) gather_max_bwd_kernel(const T* __restrict__ dout, const int32_t* __restrict__ arg,
                                                            const int64_t* __restrict__ order, const int64_t* __restrict__ offsets,
                                                            int64_t n, int k, int c, T* __restrict__ dx) {
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= n * c) return;
    const int64_t row = t / c;
    const int ch = (int)(t - row * c);
    float acc = 0.f;
    const int64_t e1 = offsets[row + 1];
    for (int64_t e = offsets[row]; e < e1; ++e) {
        const int64_t mj = order[e];
        const int64_t mm = mj / k;
        const int j = (int)(mj - mm * k);
        if (arg[mm * c + ch] == j) acc += ld1(dout + mm * c + ch);
    }
    st1(dx + t, acc);
}

}  // namespace

extern "C" {

int pps_gather_rows_f32(const float* x, const int64_t* idx, int64_t r, int c, float* out, void* stream) {
    if (r < 0 || c < 1) return 1;
    if (r == 0) return 0;
    if (!x || !idx || !out) return 1;
    if (c & 3) gather_rows_kernel<float><<<blocks_for(r * c), TB, 0, (hipStream_t)stream>>>(x, idx, r, c, out);
    else gather_rows_kernel<float4><<<blocks_for(r * (c / 4)), TB, 0, (hipStream_t)stream>>>((const float4*)x, idx, r, c / 4, (float4*)out);
    return launch_status();
}

int pps_segment_sum_rows_f32(const float* vals, const int64_t* order, const int64_t* offsets, int64_t n, int c, float* out,
                             void* stream) {
    if (n < 0 || c < 1) return 1;
    if (n == 0) return 0;
    if (!order || !offsets || !out) return 1;          /* vals may be NULL only when there are no entries at all */
    if (c & 3) segment_sum_rows_kernel<float><<<blocks_for(n * c), TB, 0, (hipStream_t)stream>>>(vals, order, offsets, n, c, out);
    else segment_sum_rows_kernel<float4><<<blocks_for(n * (c / 4)), TB, 0, (hipStream_t)stream>>>((const float4*)vals, order, offsets, n,
                                                                                              c / 4, (float4*)out);
    return launch_status();
}

int pps_segment_sum_rows_16(const void* vals, const int64_t* order, const int64_t* offsets, int64_t n, int c, int dtype, float* out, void* stream) {
    if (n < 0 || c < 4 || (c & 3) || (dtype != 1 && dtype != 2)) return 1;
    if (n == 0) return 0;
    if (!order || !offsets || !out) return 1;
    if (dtype == 2)
        segment_sum_rows_16_kernel<true><<<blocks_for(n * (c / 4)), TB, 0, (hipStream_t)stream>>>((const uint2*)vals, order, offsets, n, c / 4, (float4*)out);
    else
        segment_sum_rows_16_kernel<false><<<blocks_for(n * (c / 4)), TB, 0, (hipStream_t)stream>>>((const uint2*)vals, order, offsets, n, c / 4, (float4*)out);
    return launch_status();
}

int pps_neighbour_contract_fwd_f32(const float* x, const int64_t* idx, const float* g, int64_t m, int k, int c, float* out,
                                   void* stream) {
    if (m < 0 || k < 1 || c < 1) return 1;
    if (m == 0) return 0;
    if (!x || !idx || !g || !out) return 1;
    if ((c & 15) == 0 && k <= 16) contract_fwd_mfma_kernel<float><<<point_blocks(m), CW * 64, 0, (hipStream_t)stream>>>(x, idx, g, m, k, c, out);
    else contract_fwd_kernel<<<blocks_for(m * c), TB, 0, (hipStream_t)stream>>>(x, idx, g, m, k, c, out);
    return launch_status();
}

int pps_neighbour_contract_bwd_f32(const float* x, const int64_t* idx, const float* g, const float* dout, int64_t m, int k, int c,
                                   float* dxg, float* dg, void* stream) {
    if (m < 0 || k < 1 || c < 1) return 1;
    if (m == 0) return 0;
    if (!x || !idx || !g || !dout || (!dxg && !dg)) return 1;
    const bool mfma = (c & 15) == 0 && k <= 16;
    if (dxg) {
        if (mfma) contract_bwd_x_mfma_kernel<float><<<point_blocks(m), CW * 64, 0, (hipStream_t)stream>>>(dout, g, m, k, c, dxg);
        else contract_bwd_x_kernel<<<blocks_for(m * c), TB, 0, (hipStream_t)stream>>>(dout, g, m, k, c, dxg);
    }
    if (dg) {
        if (mfma) contract_bwd_g_mfma_kernel<float><<<point_blocks(m), CW * 64, 0, (hipStream_t)stream>>>(dout, x, idx, m, k, c, dg);
        else contract_bwd_g_kernel<<<blocks_for(m * k * KT), TB, 0, (hipStream_t)stream>>>(dout, x, idx, m, k, c, dg);
    }
    return launch_status();
}

int pps_neighbour_contract_16_supported(int k, int c) { return (c & 15) == 0 && k >= 1 && k <= 16; }

int pps_neighbour_contract_fwd(const void* x, const int64_t* idx, const float* g, int64_t m, int k, int c, int dtype, void* out, void* stream) {
    if (dtype == 0) return pps_neighbour_contract_fwd_f32((const float*)x, idx, g, m, k, c, (float*)out, stream);
    if (m < 0 || (dtype != 1 && dtype != 2) || !pps_neighbour_contract_16_supported(k, c)) return 1;
    if (m == 0) return 0;
    if (!x || !idx || !g || !out) return 1;
    if (dtype == 1) return contract_fwd_t<uint16_t>((const uint16_t*)x, idx, g, m, k, c, (uint16_t*)out, (hipStream_t)stream);
    return contract_fwd_t<half_c>((const half_c*)x, idx, g, m, k, c, (half_c*)out, (hipStream_t)stream);
}

int pps_neighbour_contract_bwd(const void* x, const int64_t* idx, const float* g, const void* dout, int64_t m, int k, int c, int dtype,
                               float* dxg, float* dg, void* stream) {
    if (dtype == 0) return pps_neighbour_contract_bwd_f32((const float*)x, idx, g, (const float*)dout, m, k, c, dxg, dg, stream);
    if (m < 0 || (dtype != 1 && dtype != 2) || !pps_neighbour_contract_16_supported(k, c)) return 1;
    if (m == 0) return 0;
    if (!x || !idx || !g || !dout || (!dxg && !dg)) return 1;
    if (dtype == 1) return contract_bwd_t<uint16_t>((const uint16_t*)x, idx, g, (const uint16_t*)dout, m, k, c, dxg, dg, (hipStream_t)stream);
    return contract_bwd_t<half_c>((const half_c*)x, idx, g, (const half_c*)dout, m, k, c, dxg, dg, (hipStream_t)stream);
}

int pps_gather_max_arg_f32(const float* x, const int64_t* idx, int64_t m, int k, int c, float* out, int32_t* arg, void* stream) {
    if (m < 0 || k < 1 || c < 1) return 1;
    if (m == 0) return 0;
    if (!x || !idx || !out || !arg) return 1;
    gather_max_arg_kernel<float><<<blocks_for(m * c), TB, 0, (hipStream_t)stream>>>(x, idx, m, k, c, out, arg);
    return launch_status();
}

int pps_gather_max_arg_16(const void* x, const int64_t* idx, int64_t m, int k, int c, int dtype, void* out, int32_t* arg, void* stream) {
    if (m < 0 || k < 1 || c < 1 || (dtype != 1 && dtype != 2)) return 1;
    if (m == 0) return 0;
    if (!x || !idx || !out || !arg) return 1;
    if (dtype == 1)
        gather_max_arg_kernel<uint16_t><<<blocks_for(m * c), TB, 0, (hipStream_t)stream>>>((const uint16_t*)x, idx, m, k, c, (uint16_t*)out, arg);
    else
        gather_max_arg_kernel<half_c><<<blocks_for(m * c), TB, 0, (hipStream_t)stream>>>((const half_c*)x, idx, m, k, c, (half_c*)out, arg);
    return launch_status();
}

int pps_gather_max_bwd_f32(const float* dout, const int32_t* arg, const int64_t* order, const int64_t* offsets, int64_t n, int k,
                           int c, float* dx, void* stream) {
    if (n < 0 || k < 1 || c < 1) return 1;
    if (n == 0) return 0;
    if (!order || !offsets || !dx) return 1;
    gather_max_bwd_kernel<float><<<blocks_for(n * c), TB, 0, (hipStream_t)stream>>>(dout, arg, order, offsets, n, k, c, dx);
    return launch_status();
}

int pps_gather_max_bwd_16(const void* dout, const int32_t* arg, const int64_t* order, const int64_t* offsets, int64_t n, int k, int c, int dtype,
                          void* dx, void* stream) {
    if (n < 0 || k < 1 || c < 1 || (dtype != 1 && dtype != 2)) return 1;
    if (n == 0) return 0;
    if (!order || !offsets || !dx) return 1;
    if (dtype == 1)
        gather_max_bwd_kernel<uint16_t><<<blocks_for(n * c), TB, 0, (hipStream_t)stream>>>((const uint16_t*)dout, arg, order, offsets, n, k, c,
                                                                                          (uint16_t*)dx);
    else
        gather_max_bwd_kernel<half_c><<<blocks_for(n * c), TB, 0, (hipStream_t)stream>>>((const half_c*)dout, arg, order, offsets, n, k, c, (half_c*)dx);
    return launch_status();
}

}  // extern "C"
