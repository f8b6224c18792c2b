// BatchNorm1d in train() mode on point-major activations [rows, C] with the activation fused, forward and backward (gfx950).
//
// replaces (reference, under autograd): every `activation(bn(conv(x)))` of source/base/nn.py (ResidualBlock :438-450,
// FKAConvNetwork :508-554, STN :162-190, PointNetfeat :323-336, MLP :376-417): torch runs collect_statistics + transform +
// relu forward and relu_backward + backward_reduce + backward_elemt backward = 7 passes over tensors of up to 1 M x 256
// elements; fused here into 2 + 2 (statistics, apply+act | mask+reduce, mask+elementwise), all pure streaming kernels:
// 4 channels per thread (8/16-byte accesses), rows strided over the block, per-channel sums reduced in a fixed order
// (per-thread fp32 partials shifted by a per-channel pivot, block partials and the final sums in double).
// fp32, bf16 or IEEE-half storage (dtype 0 / 1 / 2), fp32 arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ppsurf_amd.h"

namespace {

constexpr int BNT = 256;
constexpr int MAXBLK = 1024;

struct F4 { float v[4]; };

template <typename T> __device__ __forceinline__ F4 load4(const T* p);
template <> __device__ __forceinline__ F4 load4<float>(const float* p) {
    const float4 t = *(const float4*)p;
    return F4{{t.x, t.y, t.z, t.w}};
}
template <> __device__ __forceinline__ F4 load4<uint16_t>(const uint16_t* p) {
    const uint2 t = *(const uint2*)p;
    return F4{{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u)}};
}
typedef _Float16 half_t;
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
template <> __device__ __forceinline__ F4 load4<half_t>(const half_t* p) {
    const uint2 t = *(const uint2*)p;
    const half2v a = *(const half2v*)&t.x, b = *(const half2v*)&t.y;
    return F4{{(float)a[0], (float)a[1], (float)b[0], (float)b[1]}};
}
__device__ __forceinline__ uint32_t bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
template <typename T> __device__ __forceinline__ void store4(T* p, const F4& a);
template <> __device__ __forceinline__ void store4<float>(float* p, const F4& a) { *(float4*)p = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]); }
template <> __device__ __forceinline__ void store4<half_t>(half_t* p, const F4& a) {
    const half2v x = {(half_t)a.v[0], (half_t)a.v[1]}, y = {(half_t)a.v[2], (half_t)a.v[3]};
    *(uint2*)p = make_uint2(*(const unsigned*)&x, *(const unsigned*)&y);
}
template <> __device__ __forceinline__ void store4<uint16_t>(uint16_t* p, const F4& a) {
    *(uint2*)p = make_uint2(bf16_rne(a.v[0]) | (bf16_rne(a.v[1]) << 16), bf16_rne(a.v[2]) | (bf16_rne(a.v[3]) << 16));
}

// thread -> (channel group cg of 4 channels, row lane rl); rows rl, rl + rpb, ... of the block's slab
struct Map {
    int cg, rl, rpb, tpr;        // threads per row = C/4 (<= 256), rows per block iteration
};
__device__ __forceinline__ Map map_of(int C) {
    Map m;
    m.tpr = C >> 2;
    m.rpb = BNT / m.tpr;
    m.cg = threadIdx.x % m.tpr;
    m.rl = threadIdx.x / m.tpr;
    return m;
}

// per-block partial sums over a slab of rows: part[blk][C][2] doubles
//   MODE 0: (x - pivot), (x - pivot)^2            MODE 1: dyh, dyh * xhat   with dyh = dy * [act mask]
template <typename T, int MODE>
__global__ __launch_bounds__(BNT) void bn_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t rows, int C,
                                                       const float* __restrict__ save /* mean[C], rstd[C] (MODE 1) */,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                       double* __restrict__ part, const T* __restrict__ res = nullptr) {
    extern __shared__ double red[];                 // [rpb][C][2]
    const Map m = map_of(C);
    const int64_t slab = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * slab, r1 = r0 + slab < rows ? r0 + slab : rows;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float a[4], b[4], mu[4], rs[4];
    if (m.rl < m.rpb) {
        if (MODE == 0) {
            const F4 pv = load4<T>(x + 4 * m.cg);                       // pivot: row 0 of the tensor
#pragma unroll
            for (int i = 0; i < 4; ++i) mu[i] = pv.v[i];
            if (blockIdx.x == 0 && m.rl == 0) {
                float* pivots = (float*)(part + (int64_t)gridDim.x * 2 * C);
#pragma unroll
                for (int i = 0; i < 4; ++i) pivots[4 * m.cg + i] = pv.v[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mu[i] = save[4 * m.cg + i]; rs[i] = save[C + 4 * m.cg + i];
                a[i] = gamma[4 * m.cg + i]; b[i] = beta[4 * m.cg + i];
            }
        }
        for (int64_t r = r0 + m.rl; r < r1; r += m.rpb) {
            const F4 xv = load4<T>(x + r * C + 4 * m.cg);
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = xv.v[i] - mu[i]; s1[i] += d; s2[i] += d * d; }
            } else {
                const F4 gv = load4<T>(dy + r * C + 4 * m.cg);
                F4 rv = {{0.f, 0.f, 0.f, 0.f}};
                if (res) rv = load4<T>(res + r * C + 4 * m.cg);           // residual variant: the ReLU sits behind BN(x) + res
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xh = (xv.v[i] - mu[i]) * rs[i];
                    const float g = (relu && !(xh * a[i] + b[i] + rv.v[i] > 0.f)) ? 0.f : gv.v[i];
                    s1[i] += g; s2[i] += g * xh;
                }
            }
        }
    }
    if (m.rl < m.rpb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[((int64_t)m.rl * C + 4 * m.cg + i) * 2] = (double)s1[i];
            red[((int64_t)m.rl * C + 4 * m.cg + i) * 2 + 1] = (double)s2[i];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += BNT) {
        double s = 0.0;
        for (int r = 0; r < m.rpb; ++r) s += red[(int64_t)r * 2 * C + e];
        part[(int64_t)blockIdx.x * 2 * C + e] = s;
    }
}

// MODE 0: mean / rstd + running statistics;  MODE 1: dgamma = S2, dbeta = S1, and the means S1/n, S2/n for the elementwise pass.
// One wave per channel: lanes stride over the block partials, fixed butterfly at the end (deterministic).
template <typename T, int MODE>
__global__ __launch_bounds__(BNT) void bn_finalize_kernel(const double* __restrict__ part, int nblk, int64_t rows, int C, const T* __restrict__ x,
                                                         float eps, float momentum, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, float* __restrict__ out /* [2][C] */,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * (BNT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    // operands of the tail requested before the partial sums are walked (three dependent memory latencies less at the end of a 5 us kernel)
    float pivot_f = 0.f, rm0 = 0.f, rv0 = 0.f;
    if (MODE == 0) {
        pivot_f = ((const float*)(part + (int64_t)nblk * 2 * C))[c];          // written by block 0 of bn_reduce_kernel: reading row 0 of x here instead
                                                                              // costs 4 us (a cold page of another allocation, measured)
        if (running_mean) { rm0 = running_mean[c]; rv0 = running_var[c]; }
    }
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
    for (int b = lane; b < nblk; b += 64) {
        const double2 v = *(const double2*)(part + ((int64_t)b * C + c) * 2);
        s1 += v.x; s2 += v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane != 0) return;
    const double n = (double)rows;
    if (MODE == 0) {
        const double pivot = (double)pivot_f;
        const double d = s1 / n;
        double var = s2 / n - d * d;
        if (var < 0.0) var = 0.0;
        const double mean = pivot + d;
        out[c] = (float)mean;
        out[C + c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            running_mean[c] = (float)((1.0 - momentum) * rm0 + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * rv0 + momentum * (rows > 1 ? var * n / (n - 1.0) : var));
        }
    } else {
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        out[c] = (float)(s1 / n);
        out[C + c] = (float)(s2 / n);
    }
}

// MODE 0: y = act((x - mean) * rstd * gamma + beta);  MODE 1: dx = gamma * rstd * (dyh - m1 - xhat * m2)
// Same thread -> (channel group, row lane) map as the reductions: the per-channel coefficients live in registers.
template <typename T, int MODE>
__global__ __launch_bounds__(BNT) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t rows, int C,
                                                      const float* __restrict__ save, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ gmeans, int relu,
                                                      T* __restrict__ out, const T* __restrict__ res = nullptr, T* __restrict__ dres = nullptr) {
    const Map m = map_of(C);
    float mu[4], sc[4], sh[4], rs[4], m1[4], m2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 4 * m.cg + i;
        mu[i] = save[c]; rs[i] = save[C + c];
        sc[i] = gamma[c]; sh[i] = beta[c];
        if (MODE == 1) { m1[i] = gmeans[c]; m2[i] = gmeans[C + c]; }
    }
    for (int64_t r = (int64_t)blockIdx.x * m.rpb + m.rl; r < rows; r += (int64_t)gridDim.x * m.rpb) {
        const F4 xv = load4<T>(x + r * C + 4 * m.cg);
        F4 o;
        F4 rv = {{0.f, 0.f, 0.f, 0.f}};
        if (res) rv = load4<T>(res + r * C + 4 * m.cg);                   // residual variant: act(BN(x) + res), one pass instead of three
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float y = (xv.v[i] - mu[i]) * rs[i] * sc[i] + sh[i] + rv.v[i];
                o.v[i] = (relu && !(y > 0.f)) ? 0.f : y;
            }
        } else {
            const F4 gv = load4<T>(dy + r * C + 4 * m.cg);
            F4 gm;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (xv.v[i] - mu[i]) * rs[i];
                const float g = (relu && !(xh * sc[i] + sh[i] + rv.v[i] > 0.f)) ? 0.f : gv.v[i];
                gm.v[i] = g;
                o.v[i] = sc[i] * rs[i] * (g - m1[i] - xh * m2[i]);
            }
            if (dres) store4<T>(dres + r * C + 4 * m.cg, gm);             // the residual's gradient: the masked upstream gradient
        }
        store4<T>(out + r * C + 4 * m.cg, o);
    }
}

inline int reduce_blocks(int64_t rows, int C) {
    const int rpb = BNT / (C >> 2);
    int64_t nb = (rows + (int64_t)rpb * 8 - 1) / ((int64_t)rpb * 8);          // >= 8 iterations per thread
    if (nb < 1) nb = 1;
    return (int)(nb > MAXBLK ? MAXBLK : nb);
}
inline int apply_blocks(int64_t rows, int C) {
    const int rpb = BNT / (C >> 2);
    const int64_t nb = (rows + (int64_t)rpb * 4 - 1) / ((int64_t)rpb * 4);          // ~4 rows per thread
    return (int)(nb < 1 ? 1 : nb > 8192 ? 8192 : nb);
}
// column sums of a [rows, C] tensor (bias gradient of a row layer): per-block partials over a slab of rows, then one wave per channel
template <typename T>
__global__ __launch_bounds__(BNT) void col_sum_kernel(const T* __restrict__ x, int64_t rows, int C, int64_t ld, double* __restrict__ part) {
    extern __shared__ double red[];                 // [rpb][C]
    const Map m = map_of(C);
    const int64_t slab = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * slab, r1 = r0 + slab < rows ? r0 + slab : rows;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (m.rl < m.rpb) {
        for (int64_t r = r0 + m.rl; r < r1; r += m.rpb) {
            const F4 xv = load4<T>(x + r * ld + 4 * m.cg);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] += xv.v[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(int64_t)m.rl * C + 4 * m.cg + i] = (double)s[i];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C; e += BNT) {
        double t = 0.0;
        for (int r = 0; r < m.rpb; ++r) t += red[(int64_t)r * C + e];
        part[(int64_t)blockIdx.x * C + e] = t;
    }
}

__global__ __launch_bounds__(BNT) void col_sum_finalize_kernel(const double* __restrict__ part, int nblk, int C, float* __restrict__ out) {
    const int c = blockIdx.x * (BNT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double t = 0.0;
#pragma unroll 4
    for (int b = lane; b < nblk; b += 64) t += part[(int64_t)b * C + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) out[c] = (float)t;
}

template <typename T>
int col_sum_t(const T* x, int64_t rows, int c, int64_t ld, float* out, void* ws, hipStream_t st, int nb) {
    double* part = (double*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    const size_t lds = (size_t)(BNT / (c >> 2)) * c * sizeof(double);
    hipLaunchKernelGGL((col_sum_kernel<T>), dim3(nb), dim3(BNT), lds, st, x, rows, c, ld, part);
    hipLaunchKernelGGL(col_sum_finalize_kernel, dim3((c + 3) / 4), dim3(BNT), 0, st, (const double*)part, nb, c, out);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

inline bool ok_shape(int64_t rows, int c) { return rows >= 1 && c >= 4 && c <= 1024 && (c & 3) == 0 && (BNT % (c >> 2)) == 0; }
inline char* al(char* p) { return (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255); }

template <typename T>
int fwd_t(const T* x, int64_t rows, int c, const float* gamma, const float* beta, float* rm, float* rv, float momentum, float eps, int relu,
          T* y, float* save, void* ws, hipStream_t st, const T* res = nullptr) {
    const int nb = reduce_blocks(rows, c);
    double* part = (double*)al((char*)ws);
    const size_t lds = (size_t)(BNT / (c >> 2)) * c * 2 * sizeof(double);
    hipLaunchKernelGGL((bn_reduce_kernel<T, 0>), dim3(nb), dim3(BNT), lds, st, x, (const T*)nullptr, rows, c, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, 0, part);
    hipLaunchKernelGGL((bn_finalize_kernel<T, 0>), dim3((c + 3) / 4), dim3(BNT), 0, st, (const double*)part, nb, rows, c, x, eps, momentum,
                       rm, rv, save, (float*)nullptr, (float*)nullptr);
    const int grid = apply_blocks(rows, c);
    hipLaunchKernelGGL((bn_apply_kernel<T, 0>), dim3(grid), dim3(BNT), 0, st, x, (const T*)nullptr, rows, c, (const float*)save, gamma, beta,
                       (const float*)nullptr, relu, y, res, (T*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

template <typename T>
int bwd_t(const T* x, const T* dy, int64_t rows, int c, const float* gamma, const float* beta, const float* save, int relu, T* dx,
          float* dgamma, float* dbeta, void* ws, hipStream_t st, const T* res = nullptr, T* dres = nullptr) {
    const int nb = reduce_blocks(rows, c);
    double* part = (double*)al((char*)ws);
    float* gmeans = (float*)al((char*)(part + (size_t)nb * 2 * c));
    const size_t lds = (size_t)(BNT / (c >> 2)) * c * 2 * sizeof(double);
    hipLaunchKernelGGL((bn_reduce_kernel<T, 1>), dim3(nb), dim3(BNT), lds, st, x, dy, rows, c, save, gamma, beta, relu, part, res);
    hipLaunchKernelGGL((bn_finalize_kernel<T, 1>), dim3((c + 3) / 4), dim3(BNT), 0, st, (const double*)part, nb, rows, c, x, 0.f, 0.f,
                       (float*)nullptr, (float*)nullptr, gmeans, dgamma, dbeta);
    const int grid = apply_blocks(rows, c);
    hipLaunchKernelGGL((bn_apply_kernel<T, 1>), dim3(grid), dim3(BNT), 0, st, x, dy, rows, c, save, gamma, beta, (const float*)gmeans, relu, dx, res, dres);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace

extern "C" {

size_t pps_bn_train_ws_bytes(int64_t rows, int c) {
    if (!ok_shape(rows, c)) return 0;
    return 1024 + (size_t)reduce_blocks(rows, c) * 2 * c * sizeof(double) + 4 * (size_t)c * sizeof(float);        // partials | pivots or gradient means
}

int pps_col_sum_strided(const void* x, int64_t rows, int c, int64_t ld, int dtype, float* out, void* ws, void* stream) {
    if (!ok_shape(rows, c) || (dtype < 0 || dtype > 2) || !x || !out || !ws || ld < c || (ld & 3) || ((uintptr_t)x & (dtype == 0 ? 15 : 7))) return 1;
    int nb = reduce_blocks(rows, c);
    if (nb > 256) nb = 256;                              // one pass of <= 256 partial rows per channel in the second kernel
    if (dtype == 0) return col_sum_t<float>((const float*)x, rows, c, ld, out, ws, (hipStream_t)stream, nb);
    if (dtype == 2) return col_sum_t<half_t>((const half_t*)x, rows, c, ld, out, ws, (hipStream_t)stream, nb);
    return col_sum_t<uint16_t>((const uint16_t*)x, rows, c, ld, out, ws, (hipStream_t)stream, nb);
}

int pps_col_sum(const void* x, int64_t rows, int c, int dtype, float* out, void* ws, void* stream) {
    return pps_col_sum_strided(x, rows, c, (int64_t)c, dtype, out, ws, stream);
}

int pps_bn_train_fwd(const void* x, int64_t rows, int c, int dtype, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float momentum, float eps, int relu, void* y, float* save, void* ws, void* stream) {
    if (rows == 0) return 0;
    if (!ok_shape(rows, c) || (dtype < 0 || dtype > 2) || !x || !gamma || !beta || !y || !save || !ws) return 1;
    if ((running_mean == nullptr) != (running_var == nullptr)) return 1;
    if (dtype == 0)
        return fwd_t<float>((const float*)x, rows, c, gamma, beta, running_mean, running_var, momentum, eps, relu, (float*)y, save, ws, (hipStream_t)stream);
    if (dtype == 2)
        return fwd_t<half_t>((const half_t*)x, rows, c, gamma, beta, running_mean, running_var, momentum, eps, relu, (half_t*)y, save, ws, (hipStream_t)stream);
    return fwd_t<uint16_t>((const uint16_t*)x, rows, c, gamma, beta, running_mean, running_var, momentum, eps, relu, (uint16_t*)y, save, ws,
                           (hipStream_t)stream);
}

int pps_bn_train_bwd(const void* x, const void* dy, int64_t rows, int c, int dtype, const float* gamma, const float* beta, const float* save,
                     int relu, void* dx, float* dgamma, float* dbeta, void* ws, void* stream) {
    if (rows == 0) return 0;
    if (!ok_shape(rows, c) || (dtype < 0 || dtype > 2) || !x || !dy || !gamma || !beta || !save || !dx || !dgamma || !dbeta || !ws) return 1;
    if (dtype == 0)
        return bwd_t<float>((const float*)x, (const float*)dy, rows, c, gamma, beta, save, relu, (float*)dx, dgamma, dbeta, ws, (hipStream_t)stream);
    if (dtype == 2)
        return bwd_t<half_t>((const half_t*)x, (const half_t*)dy, rows, c, gamma, beta, save, relu, (half_t*)dx, dgamma, dbeta, ws, (hipStream_t)stream);
    return bwd_t<uint16_t>((const uint16_t*)x, (const uint16_t*)dy, rows, c, gamma, beta, save, relu, (uint16_t*)dx, dgamma, dbeta, ws,
                           (hipStream_t)stream);
}

/* relu(BN(x) + res) in train() mode: the tail of a residual block (source/base/nn.py:448-450 `self.activation(x + shortcut)` behind bn2) as the
 * BatchNorm's own apply pass -- res [rows, c] in the storage type of x.  Backward: dx as pps_bn_train_bwd with the ReLU mask taken behind the sum,
 * dres [rows, c] = the masked upstream gradient (the shortcut's gradient). */
int pps_bn_add_relu_fwd(const void* x, const void* res, int64_t rows, int c, int dtype, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float momentum, float eps, void* y, float* save, void* ws, void* stream) {
    if (rows == 0) return 0;
    if (!ok_shape(rows, c) || (dtype < 0 || dtype > 2) || !x || !res || !gamma || !beta || !y || !save || !ws) return 1;
    if ((running_mean == nullptr) != (running_var == nullptr)) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) return fwd_t<float>((const float*)x, rows, c, gamma, beta, running_mean, running_var, momentum, eps, 1, (float*)y, save, ws, st, (const float*)res);
    if (dtype == 2)
        return fwd_t<half_t>((const half_t*)x, rows, c, gamma, beta, running_mean, running_var, momentum, eps, 1, (half_t*)y, save, ws, st, (const half_t*)res);
    return fwd_t<uint16_t>((const uint16_t*)x, rows, c, gamma, beta, running_mean, running_var, momentum, eps, 1, (uint16_t*)y, save, ws, st, (const uint16_t*)res);
}

int pps_bn_add_relu_bwd(const void* x, const void* res, const void* dy, int64_t rows, int c, int dtype, const float* gamma, const float* beta,
                        const float* save, void* dx, void* dres, float* dgamma, float* dbeta, void* ws, void* stream) {
    if (rows == 0) return 0;
    if (!ok_shape(rows, c) || (dtype < 0 || dtype > 2) || !x || !res || !dy || !gamma || !beta || !save || !dx || !dres || !dgamma || !dbeta || !ws) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        return bwd_t<float>((const float*)x, (const float*)dy, rows, c, gamma, beta, save, 1, (float*)dx, dgamma, dbeta, ws, st, (const float*)res, (float*)dres);
    if (dtype == 2)
        return bwd_t<half_t>((const half_t*)x, (const half_t*)dy, rows, c, gamma, beta, save, 1, (half_t*)dx, dgamma, dbeta, ws, st, (const half_t*)res,
                             (half_t*)dres);
    return bwd_t<uint16_t>((const uint16_t*)x, (const uint16_t*)dy, rows, c, gamma, beta, save, 1, (uint16_t*)dx, dgamma, dbeta, ws, st,
                           (const uint16_t*)res, (uint16_t*)dres);
}

}  // extern "C"
