// Attention pooling of the interpolation head in the TRAINING step, forward and backward (gfx950).
//
// replaces (reference, under autograd): source/poco_model.py:412-414
//     attention_weights = softmax(query, dim = neighbours).mean(dim = heads);   x = matmul(attention_weights, value)
// in the pooled form of DESIGN.md section 2 (identity 2: fc_value runs after the pooling): per query q with k <= 64 neighbours,
// H <= 64 heads (64 in the interpolation head; 1 in PointNet's attention pooling, source/base/nn.py:84-96) and C <= 256 channels
//     a[j] = 1/H * sum_h softmax_j(qy[q, j, h]),     pooled[q, c] = sum_j a[j] * h[q, j, c].
// torch runs this as softmax (fwd + bwd), mean, cast, two batched GEMMs and their transposes over [Q*k, H] and [Q*k, C] tensors
// (~3 ms per step at Q*k = 1.28 M); here it is ONE streaming kernel each way: a workgroup per query keeps the 64 x 64 logits in LDS,
// reads the neighbour rows once, and writes only the pooled row (forward) or the two gradients (backward; the softmax is recomputed
// from the logits, nothing but the inputs is saved).  HBM-bound: 40 KB (fwd) / 80 KB (bwd) per query in bf16.
// Storage type T: float or bf16 (as uint16), arithmetic fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pps_common.h"
#include "../../include/ppsurf_amd.h"

#define PPS_OK 0
#define PPS_ERR_ARG 1
#define PPS_ERR_LAUNCH 2

namespace {

constexpr int AT_NT = 256;      // threads per workgroup = 4 waves; wave w owns neighbours [16w, 16w+16), lane = head
constexpr int AT_H = 64;
constexpr int AT_KMAX = 64;

typedef _Float16 half_t;                       // IEEE half storage (trainer.precision 16-mixed); uint16_t stands for bfloat16
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ld(const half_t* p, int64_t i) { return (float)p[i]; }
__device__ __forceinline__ void st(half_t* p, int64_t i, float v) { p[i] = (half_t)v; }
__device__ __forceinline__ float ld(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float ld(const uint16_t* p, int64_t i) { return __uint_as_float((unsigned)p[i] << 16); }
__device__ __forceinline__ void st(float* p, int64_t i, float v) { p[i] = v; }
__device__ __forceinline__ void st(uint16_t* p, int64_t i, float v) {            // round to nearest even, like torch's float -> bfloat16
    unsigned u = __float_as_uint(v);
    if ((u & 0x7f800000u) == 0x7f800000u && (u & 0x007fffffu)) { p[i] = (uint16_t)((u >> 16) | 0x40); return; }
    u += 0x7fffu + ((u >> 16) & 1u);
    p[i] = (uint16_t)(u >> 16);
}

// four consecutive channels of one row per lane (8 bytes in bf16, 16 in fp32): a wave covers a 256-channel row with ONE load
__device__ __forceinline__ float4 ld4(const float* p, int64_t i) { return *(const float4*)(p + i); }
__device__ __forceinline__ float4 ld4(const uint16_t* p, int64_t i) {
    const uint2 u = *(const uint2*)(p + i);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ float4 ld4(const half_t* p, int64_t i) {
    const uint2 u = *(const uint2*)(p + i);
    const half2v a = *(const half2v*)&u.x, b = *(const half2v*)&u.y;
    return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
}
__device__ __forceinline__ void st4(half_t* p, int64_t i, float4 v) {
    const half2v a = {(half_t)v.x, (half_t)v.y}, b = {(half_t)v.z, (half_t)v.w};
    *(uint2*)(p + i) = make_uint2(*(const unsigned*)&a, *(const unsigned*)&b);
}
__device__ __forceinline__ void st4(float* p, int64_t i, float4 v) { *(float4*)(p + i) = v; }
__device__ __forceinline__ void st4(uint16_t* p, int64_t i, float4 v) {
    uint16_t t[4];
    st(t, 0, v.x); st(t, 1, v.y); st(t, 2, v.z); st(t, 3, v.w);
    *(uint2*)(p + i) = make_uint2((unsigned)t[0] | ((unsigned)t[1] << 16), (unsigned)t[2] | ((unsigned)t[3] << 16));
}

// reductions over the 64 lanes, every lane gets the result: DPP row reduction (4 moves), then the row and half swaps of gfx950
// (pps_common.h lane_xor_u32) instead of six ds_bpermute round trips -- these kernels reduce once per neighbour row
__device__ __forceinline__ float lane_xor_f32(float v, int st) {
    return __uint_as_float(pps::lane_xor_u32(__float_as_uint(v), st, (int)(threadIdx.x & 63)));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = pps::row16_sum(v);
    v += lane_xor_f32(v, 16);
    v += lane_xor_f32(v, 32);
    return v;
}

// logits of query q -> LDS e[j][h] = exp(qy - max_j), inv_s[h] = 1 / sum_j e; returns nothing, all threads sync'ed on exit
template <typename T>
__device__ __forceinline__ void softmax_to_lds(const T* __restrict__ qy, int64_t q, int k, int H, float (*e)[AT_H + 1], float* red, float* inv_s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* src = qy + q * (int64_t)k * H;
    for (int i = threadIdx.x; i < k * AT_H; i += AT_NT) {                            // heads beyond H: logit 0 (their columns are never used)
        const int j = i >> 6, hh = i & 63;
        e[j][hh] = hh < H ? ld(src, (int64_t)j * H + hh) : 0.f;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = wave * 16; j < wave * 16 + 16 && j < k; ++j) m = fmaxf(m, e[j][lane]);
    red[wave * 64 + lane] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[lane], red[64 + lane]), fmaxf(red[128 + lane], red[192 + lane]));
    __syncthreads();
    float s = 0.f;
    for (int j = wave * 16; j < wave * 16 + 16 && j < k; ++j) {
        const float v = __expf(e[j][lane] - m);
        e[j][lane] = v;
        s += v;
    }
    red[wave * 64 + lane] = s;
    __syncthreads();
    if (threadIdx.x < 64) inv_s[lane] = 1.f / (red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane]);
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(AT_NT) void attn_pool_fwd_kernel(const T* __restrict__ qy, const T* __restrict__ h, int64_t Q, int k, int H, int C,
                                                              int relu_h, T* __restrict__ pooled) {
    const float hfloor = relu_h ? 0.f : -INFINITY;                                  // relu_h: h is stored BEFORE its ReLU (train_ops.Act)
    __shared__ float e[AT_KMAX][AT_H + 1];
    __shared__ float red[4 * 64], inv_s[64], a[AT_KMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t q = blockIdx.x; q < Q; q += gridDim.x) {
        softmax_to_lds(qy, q, k, H, e, red, inv_s);
        const float inv_h = 1.f / (float)H;
        for (int j = wave * 16; j < wave * 16 + 16 && j < k; ++j) {                  // a[j] = mean over the heads
            const float v = wave_sum(lane < H ? e[j][lane] * inv_s[lane] : 0.f);
            if (lane == 0) a[j] = v * inv_h;
        }
        __syncthreads();
        const T* hq = h + q * (int64_t)k * C;
        if (C == 256) {                                                              // wave w sums its 16 rows, lane = 4 channels
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j0 = wave * 16; j0 < wave * 16 + 16 && j0 < k; j0 += 4) {       // four rows in flight per wave
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = ld4(hq, (int64_t)(j0 + u < k ? j0 + u : k - 1) * 256 + 4 * lane);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float aj = j0 + u < k ? a[j0 + u] : 0.f;
                    acc.x += aj * fmaxf(v[u].x, hfloor); acc.y += aj * fmaxf(v[u].y, hfloor);
                    acc.z += aj * fmaxf(v[u].z, hfloor); acc.w += aj * fmaxf(v[u].w, hfloor);
                }
            }
            float4* part = (float4*)&e[0][0];                                        // the logits are no longer needed: 4 x 64 float4
            __syncthreads();
            part[wave * 64 + lane] = acc;
            __syncthreads();
            if (wave == 0) {
                const float4 p0 = part[lane], p1 = part[64 + lane], p2 = part[128 + lane], p3 = part[192 + lane];
                st4(pooled, q * 256 + 4 * lane, make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y),
                                                            (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w)));
            }
        } else {
            for (int c = threadIdx.x; c < C; c += AT_NT) {
                float acc = 0.f;
                for (int j = 0; j < k; ++j) acc += a[j] * fmaxf(ld(hq, (int64_t)j * C + c), hfloor);
                st(pooled, q * (int64_t)C + c, acc);
            }
        }
        __syncthreads();
    }
}

// d_h[q,j,c] = a[j] dP[c];   da[j] = sum_c dP[c] h[q,j,c];   d_qy[q,j,h] = s[j,h]/H * (da[j] - sum_j' s[j',h] da[j'])
// dh == NULL: d_h is not stored -- the caller gets the weights a [Q, k] (a_out) and rebuilds mask * a[j] * dP[c] where it consumes the gradient
// (the input-gradient kernel of fc_query, LayerArgs::att_a): [Q k, C] written here and read back there otherwise, for one multiply per element
template <typename T>
__global__ __launch_bounds__(AT_NT) void attn_pool_bwd_kernel(const T* __restrict__ qy, const T* __restrict__ h, const T* __restrict__ dpooled,
                                                              int64_t Q, int k, int H, int C, int relu_h, T* __restrict__ dqy, T* __restrict__ dh,
                                                              float* __restrict__ a_out) {
    const float hfloor = relu_h ? 0.f : -INFINITY;
    __shared__ float e[AT_KMAX][AT_H + 1];
    __shared__ float red[4 * 64], inv_s[64], a[AT_KMAX], da[AT_KMAX], dp[256], dsum[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t q = blockIdx.x; q < Q; q += gridDim.x) {
        softmax_to_lds(qy, q, k, H, e, red, inv_s);
        const float inv_h = 1.f / (float)H;
        for (int c = threadIdx.x; c < C; c += AT_NT) dp[c] = ld(dpooled, q * (int64_t)C + c);
        for (int j = wave * 16; j < wave * 16 + 16 && j < k; ++j) {
            const float s = lane < H ? e[j][lane] * inv_s[lane] : 0.f;
            e[j][lane] = s;                                                          // e now holds the softmax probabilities s[j][h]
            const float v = wave_sum(s);
            if (lane == 0) {
                a[j] = v * inv_h;
                if (a_out) a_out[q * (int64_t)k + j] = v * inv_h;
            }
        }
        __syncthreads();
        const T* hq = h + q * (int64_t)k * C;
        T* dhq = dh ? dh + q * (int64_t)k * C : nullptr;
        if (C == 256) {
            const float4 d4 = make_float4(dp[4 * lane], dp[4 * lane + 1], dp[4 * lane + 2], dp[4 * lane + 3]);
            // one neighbour row per wave pass, 4 channels per lane; the wave's 16 rows are requested four at a time (a row per iteration left
            // the loads of a row waiting behind the shuffles of the previous one)
            for (int j0 = wave * 16; j0 < wave * 16 + 16 && j0 < k; j0 += 4) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = ld4(hq, (int64_t)(j0 + u < k ? j0 + u : k - 1) * 256 + 4 * lane);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u;
                    if (j < k) {
                        const float aj = a[j];
                        // with relu_h the gradient goes to the stored pre-activation: masked where it was clipped
                        if (dhq)
                            st4(dhq, (int64_t)j * 256 + 4 * lane, make_float4(v[u].x > hfloor ? aj * d4.x : 0.f, v[u].y > hfloor ? aj * d4.y : 0.f,
                                                                              v[u].z > hfloor ? aj * d4.z : 0.f, v[u].w > hfloor ? aj * d4.w : 0.f));
                        const float4 w = make_float4(fmaxf(v[u].x, hfloor), fmaxf(v[u].y, hfloor), fmaxf(v[u].z, hfloor), fmaxf(v[u].w, hfloor));
                        const float part = wave_sum((d4.x * w.x + d4.y * w.y) + (d4.z * w.z + d4.w * w.w));
                        if (lane == 0) da[j] = part;
                    }
                }
            }
        } else {
            for (int j = wave * 16; j < wave * 16 + 16 && j < k; ++j) {
                float part = 0.f;
                const float aj = a[j];
                for (int c = lane; c < C; c += 64) {
                    const float hv = ld(hq, (int64_t)j * C + c);
                    part += dp[c] * fmaxf(hv, hfloor);
                    if (dhq) st(dhq, (int64_t)j * C + c, hv > hfloor ? aj * dp[c] : 0.f);
                }
                part = wave_sum(part);
                if (lane == 0) da[j] = part;
            }
        }
        __syncthreads();
        float d = 0.f;                                                                // D[h] = sum_j s[j][h] da[j]
        for (int j = wave * 16; j < wave * 16 + 16 && j < k; ++j) d += e[j][lane] * da[j];
        red[wave * 64 + lane] = d;
        __syncthreads();
        if (threadIdx.x < 64) dsum[lane] = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
        __syncthreads();
        T* dq = dqy + q * (int64_t)k * H;
        for (int i = threadIdx.x; i < k * H; i += AT_NT) {
            const int j = i / H, hh = i - j * H;
            st(dq, i, e[j][hh] * inv_h * (da[j] - dsum[hh]));
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-head attention pooling over the k <= 64 rows of a group with the LOGIT computed inside (PointNet's AttentionPoco,
// source/base/nn.py:84-96, on the raw output of conv3: the logit fc_query(bn3(y)) = y . (w_q * scale) + const and softmax ignores the constant):
//     a = softmax_j(h_j . v),   pooled = sum_j a_j h_j            h [Q, k, 256] bf16, v [256] fp32, pooled [Q, 256] fp32
// One wave per group, 4 channels per lane; the per-row dot products of a group are turned into one value per lane by recursive halving.
// Backward: dh_j = a_j dP + dl_j v,  dl_j = a_j (dP . h_j - sum_i a_i dP . h_i),  dv = sum_{q,j} dl_j h_j  (per-wave partials, summed by the caller)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PA_K = 64;
constexpr int PA_WAVES = 4;
constexpr int PA_MAXBLK = 2048;      // 8 workgroups per CU: the kernels alternate load and compute phases per group, more resident waves hide the loads

__device__ __forceinline__ float4 unpack4(const uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ float wave_max(float v) {
    v = pps::row16_max(v);
    v = fmaxf(v, lane_xor_f32(v, 16));
    v = fmaxf(v, lane_xor_f32(v, 32));
    return v;
}

// 32 per-lane partial sums (one per row of a half patch) -> lanes l and l + 32 hold the total of row l: recursive halving within each half of
// the wave (31 exchanges instead of 32 x 6), then the two halves are added.  The patch is reduced as two half patches of 32 rows so that only
// 32 (forward) / 64 (backward) partial sums are live at a time (61 instead of 91 VGPRs forward, 166 instead of 172 backward; measured: the
// extra resident waves change nothing, neither kernel is bound by occupancy).
// (template recursion: every array index is a compile-time constant, the array stays in registers)
template <int HALF>
__device__ __forceinline__ void halve_rows(float (&p)[32], int lane) {
    const bool up = (lane & HALF) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const float send = up ? p[i] : p[i + HALF];
        const float keep = up ? p[i + HALF] : p[i];
        p[i] = keep + lane_xor_f32(send, HALF);
    }
    if constexpr (HALF > 1) halve_rows<HALF / 2>(p, lane);
}
__device__ __forceinline__ float rows_to_lanes32(float (&p)[32], int lane) {
    halve_rows<16>(p, lane);
    return p[0] + lane_xor_f32(p[0], 32);
}

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

template <typename T>
__device__ __forceinline__ float4 row4(const T* __restrict__ src, int j, int lane) { return ld4(src, (int64_t)j * 256 + 4 * lane); }

// lane j: softmax weight of row j (0 beyond k).  The rows are read here for the logits and AGAIN by the caller for the pooling (the
// 25 KB of a group come from L2 the second time): keeping them in registers costs more in occupancy than the second read.
// KP: rows actually loaded (k <= KP <= 64, a compile-time bound of 16 / 32 / 50 / 64 rows): the re-reads of the last row that fill up to KP all
// hit ONE address from one wave and serialise in the memory pipeline -- 14 of them (k = 50 under the old fixed bound of 64) cost a quarter of the kernel
template <typename T, int KP>
__device__ __forceinline__ float patch_softmax(const T* __restrict__ src, int k, const float4 v4, int lane) {
    float d[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float p[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {                             // no branch around the loads: rows beyond k re-read the last row and are zeroed
            const int j = 32 * half + i;
            if (j < KP) {
                const float t = dot4(row4(src, j < k ? j : k - 1, lane), v4);
                p[i] = j < k ? t : 0.f;
            } else {
                p[i] = 0.f;
            }
        }
        d[half] = (32 * half < KP) ? rows_to_lanes32(p, lane) : 0.f;
    }
    const float dd = lane < 32 ? d[0] : d[1];                      // lane j: logit of row j
    const float logit = lane < k ? dd : -INFINITY;
    const float m = wave_max(logit);
    const float e = lane < k ? __expf(logit - m) : 0.f;
    return e / wave_sum(e);
}

template <typename T, int KP>
__global__ __launch_bounds__(PA_WAVES * 64, 2) void patch_attn_fwd_kernel(const T* __restrict__ h, const float* __restrict__ v, int64_t Q, int k,
                                                                         float* __restrict__ pooled) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 v4 = *(const float4*)(v + 4 * lane);
    for (int64_t q = (int64_t)blockIdx.x * PA_WAVES + wave; q < Q; q += (int64_t)gridDim.x * PA_WAVES) {
        const T* src = h + q * (int64_t)k * 256;
        const float a = patch_softmax<T, KP>(src, k, v4, lane);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int j = 0; j < k; ++j) {
            const float aj = __shfl(a, j);
            const float4 x = row4(src, j, lane);
            acc.x += aj * x.x; acc.y += aj * x.y; acc.z += aj * x.z; acc.w += aj * x.w;
        }
        *(float4*)(pooled + q * 256 + 4 * lane) = acc;
    }
}

template <typename T, int KP>
__global__ __launch_bounds__(PA_WAVES * 64, 2) void patch_attn_bwd_kernel(const T* __restrict__ h, const float* __restrict__ v,
                                                                         const float* __restrict__ dpooled, int64_t Q, int k, T* __restrict__ dh,
                                                                         float* __restrict__ dv_part, float* __restrict__ a_out = nullptr,
                                                                         float* __restrict__ dl_out = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 v4 = *(const float4*)(v + 4 * lane);
    float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t q = (int64_t)blockIdx.x * PA_WAVES + wave; q < Q; q += (int64_t)gridDim.x * PA_WAVES) {
        const T* src = h + q * (int64_t)k * 256;
        const float4 dp = *(const float4*)(dpooled + q * 256 + 4 * lane);
        float dh2[2], th2[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float p[32], pt[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int j = 32 * half + i;
                if (j < KP) {
                    const float4 x = row4(src, j < k ? j : k - 1, lane);      // no branch around the loads
                    p[i] = j < k ? dot4(x, v4) : 0.f;
                    pt[i] = j < k ? dot4(x, dp) : 0.f;
                } else {
                    p[i] = pt[i] = 0.f;
                }
            }
            dh2[half] = (32 * half < KP) ? rows_to_lanes32(p, lane) : 0.f;
            th2[half] = (32 * half < KP) ? rows_to_lanes32(pt, lane) : 0.f;
        }
        const float d = lane < 32 ? dh2[0] : dh2[1];
        const float t = lane < 32 ? th2[0] : th2[1];             // lane j: dP . h_j
        const float logit = lane < k ? d : -INFINITY;
        const float m = wave_max(logit);
        const float e = lane < k ? __expf(logit - m) : 0.f;
        const float a = e / wave_sum(e);
        const float tbar = wave_sum(a * t);
        const float dl = a * (t - tbar);
        // dh == NULL: dh[q, j, :] = a_j dP + dl_j v has rank two per group -- the caller gets (a, dl) per row and the consumers of dh rebuild it on
        // load (pps_rows_layer_bwd_rank2); the [Q k, 256] tensor is neither written here nor read there
        if (a_out && lane < k) {
            a_out[q * (int64_t)k + lane] = a;
            dl_out[q * (int64_t)k + lane] = dl;
        }
        T* dst = dh ? dh + q * (int64_t)k * 256 + 4 * lane : nullptr;
#pragma unroll 8
        for (int j = 0; j < k; ++j) {
            const float aj = __shfl(a, j), dj = __shfl(dl, j);
            const float4 x = row4(src, j, lane);
            if (dst) st4(dst, (int64_t)j * 256, make_float4(aj * dp.x + dj * v4.x, aj * dp.y + dj * v4.y, aj * dp.z + dj * v4.z, aj * dp.w + dj * v4.w));
            dv.x += dj * x.x; dv.y += dj * x.y; dv.z += dj * x.z; dv.w += dj * x.w;
        }
    }
    *(float4*)(dv_part + ((int64_t)blockIdx.x * PA_WAVES + wave) * 256 + 4 * lane) = dv;
}

int patch_grid(int64_t q) {
    const int64_t blocks = (q + PA_WAVES - 1) / PA_WAVES;
    return (int)(blocks < PA_MAXBLK ? blocks : PA_MAXBLK);
}

int grid_for(int64_t q) {
    const int64_t cap = 256 * 16;
    return (int)(q < cap ? q : cap);
}

}  // namespace

extern "C" {

int pps_attn_pool_fwd(const void* qy, const void* h, int64_t q, int k, int heads, int c, int bf16, int relu_h, void* pooled, void* stream) {
    if (q < 0 || k < 1 || k > AT_KMAX || heads < 1 || heads > AT_H || c < 1 || c > 256) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!qy || !h || !pooled) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (bf16 == 2)
        hipLaunchKernelGGL(attn_pool_fwd_kernel<half_t>, dim3(grid_for(q)), dim3(AT_NT), 0, st, (const half_t*)qy, (const half_t*)h, q, k, heads, c, relu_h,
                           (half_t*)pooled);
    else if (bf16)
        hipLaunchKernelGGL(attn_pool_fwd_kernel<uint16_t>, dim3(grid_for(q)), dim3(AT_NT), 0, st, (const uint16_t*)qy, (const uint16_t*)h, q, k, heads, c,
                           relu_h, (uint16_t*)pooled);
    else
        hipLaunchKernelGGL(attn_pool_fwd_kernel<float>, dim3(grid_for(q)), dim3(AT_NT), 0, st, (const float*)qy, (const float*)h, q, k, heads, c,
                           relu_h, (float*)pooled);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

static int attn_pool_bwd_any(const void* qy, const void* h, const void* dpooled, int64_t q, int k, int heads, int c, int bf16, int relu_h, void* dqy,
                             void* dh, float* a_out, void* stream) {
    if (q < 0 || k < 1 || k > AT_KMAX || heads < 1 || heads > AT_H || c < 1 || c > 256) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!qy || !h || !dpooled || !dqy || (!dh && !a_out)) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (bf16 == 2)
        hipLaunchKernelGGL(attn_pool_bwd_kernel<half_t>, dim3(grid_for(q)), dim3(AT_NT), 0, st, (const half_t*)qy, (const half_t*)h, (const half_t*)dpooled, q,
                           k, heads, c, relu_h, (half_t*)dqy, (half_t*)dh, a_out);
    else if (bf16)
        hipLaunchKernelGGL(attn_pool_bwd_kernel<uint16_t>, dim3(grid_for(q)), dim3(AT_NT), 0, st, (const uint16_t*)qy, (const uint16_t*)h,
                           (const uint16_t*)dpooled, q, k, heads, c, relu_h, (uint16_t*)dqy, (uint16_t*)dh, a_out);
    else
        hipLaunchKernelGGL(attn_pool_bwd_kernel<float>, dim3(grid_for(q)), dim3(AT_NT), 0, st, (const float*)qy, (const float*)h,
                           (const float*)dpooled, q, k, heads, c, relu_h, (float*)dqy, (float*)dh, a_out);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_attn_pool_bwd(const void* qy, const void* h, const void* dpooled, int64_t q, int k, int heads, int c, int bf16, int relu_h, void* dqy,
                      void* dh, void* stream) {
    if (q > 0 && !dh) return PPS_ERR_ARG;
    return attn_pool_bwd_any(qy, h, dpooled, q, k, heads, c, bf16, relu_h, dqy, dh, nullptr, stream);
}

int pps_attn_pool_bwd_weights(const void* qy, const void* h, const void* dpooled, int64_t q, int k, int heads, int c, int bf16, int relu_h, void* dqy,
                              float* weights, void* stream) {
    if (q > 0 && !weights) return PPS_ERR_ARG;
    return attn_pool_bwd_any(qy, h, dpooled, q, k, heads, c, bf16, relu_h, dqy, nullptr, weights, stream);
}

/* Single-head attention pooling with the logit computed inside: a = softmax_j(h[q,j,:] . v), pooled[q,:] = sum_j a_j h[q,j,:]
 * (h [q, k, 256] bfloat16, k <= 64, v [256], pooled [q, 256] fp32).  Backward: dh [q, k, 256] bfloat16 and dv_part
 * [pps_patch_attn_partials(q)][256] whose sum over the first axis is dv. */
int pps_patch_attn_partials(int64_t q) { return q > 0 ? patch_grid(q) * PA_WAVES : 0; }

int pps_patch_attn_fwd(const void* h, const float* v, int64_t q, int k, int c, int dtype, float* pooled, void* stream) {
    if (q < 0 || k < 1 || k > PA_K || c != 256 || (dtype != 1 && dtype != 2)) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!h || !v || !pooled) return PPS_ERR_ARG;
    const dim3 grid(patch_grid(q)), block(PA_WAVES * 64);
    hipStream_t st = (hipStream_t)stream;
#define PPS_PA_FWD(T, KP) hipLaunchKernelGGL((patch_attn_fwd_kernel<T, KP>), grid, block, 0, st, (const T*)h, v, q, k, pooled)
#define PPS_PA_BY_K(CALL, T) do { if (k <= 16) CALL(T, 16); else if (k <= 32) CALL(T, 32); else if (k <= 50) CALL(T, 50); else CALL(T, 64); } while (0)
    if (dtype == 2) PPS_PA_BY_K(PPS_PA_FWD, half_t);
    else PPS_PA_BY_K(PPS_PA_FWD, uint16_t);
#undef PPS_PA_FWD
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

static int patch_attn_bwd_any(const void* h, const float* v, const float* dpooled, int64_t q, int k, int c, int dtype, void* dh, float* dv_part, float* a_out,
                              float* dl_out, void* stream) {
    if (q < 0 || k < 1 || k > PA_K || c != 256 || (dtype != 1 && dtype != 2)) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!h || !v || !dpooled || !dv_part || (!dh && !(a_out && dl_out))) return PPS_ERR_ARG;
    const dim3 grid(patch_grid(q)), block(PA_WAVES * 64);
    hipStream_t st = (hipStream_t)stream;
#define PPS_PA_BWD(T, KP) hipLaunchKernelGGL((patch_attn_bwd_kernel<T, KP>), grid, block, 0, st, (const T*)h, v, dpooled, q, k, (T*)dh, dv_part, a_out, dl_out)
    if (dtype == 2) PPS_PA_BY_K(PPS_PA_BWD, half_t);
    else PPS_PA_BY_K(PPS_PA_BWD, uint16_t);
#undef PPS_PA_BWD
#undef PPS_PA_BY_K
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_patch_attn_bwd(const void* h, const float* v, const float* dpooled, int64_t q, int k, int c, int dtype, void* dh, float* dv_part, void* stream) {
    if (q > 0 && !dh) return PPS_ERR_ARG;
    return patch_attn_bwd_any(h, v, dpooled, q, k, c, dtype, dh, dv_part, nullptr, nullptr, stream);
}

int pps_patch_attn_bwd_weights(const void* h, const float* v, const float* dpooled, int64_t q, int k, int c, int dtype, float* weights, float* dlogits,
                               float* dv_part, void* stream) {
    if (q > 0 && (!weights || !dlogits)) return PPS_ERR_ARG;
    return patch_attn_bwd_any(h, v, dpooled, q, k, c, dtype, nullptr, dv_part, weights, dlogits, stream);
}

}  // extern "C"
