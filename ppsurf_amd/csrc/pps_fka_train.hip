// FKAConv geometry branch for TRAINING: forward in train() mode and hand-written backward (gfx950).
//
// replaces (reference, under autograd): source/base/nn.py:601-643 -- neighbour offsets, norm_radius EMA (:608-613), distance
// weights, fc1 -> InstanceNorm -> act -> weighted max-pool -> fc2 -> InstanceNorm -> act -> max-pool -> fc3 -> act * dw,
// i.e. everything of FKAConvLayer.forward that produces the [M, K, 16] kernel-weighting matrix `g`; ~60 ATen launches
// forward and ~150 backward per layer on [B, M, K, 16] tensors (12 ms per 10 x 10k-point layer) become 4 + 3 launches.
//
// Mapping as in the inference kernels: 16 lanes (one DPP row) hold the K <= 16 neighbours of one support point, each lane
// its neighbour's 16 channels in registers; the per-layer parameters (1140 floats, `geo`) are read with scalar loads.
// Nothing but `g` is stored by the forward pass: the backward pass RECOMPUTES the branch in each of its three passes (the
// two InstanceNorms are reductions over all (point, neighbour) pairs of a shape, so their backward needs the sums
// sum(dy), sum(dy * xhat) of a whole shape before the gradient can go further up):
//   pass 1: dg -> fc3, pool 2, act        -> dy2 (kept in scratch), sums for IN2, dW3
//   pass 2: IN2 backward -> fc2, pool 1   -> dy1 (overwrites dy2),  sums for IN1, dW2
//   pass 3: IN1 backward -> fc1           -> dW1; distance-weight backward -> d alpha, d beta
// Weight gradients are outer products summed over all lanes: per tile the wave stages (dz, input) in LDS and accumulates
// dz^T . input with fp32 MFMA (v_mfma_f32_16x16x4_f32) in registers across its tiles; per-block partials are then added in a
// fixed order (deterministic).  Statistics and their gradients are reduced in double.
// gradients and train-mode forward values are compared with tolerances, not bit for bit: let the compiler fuse a*b+c in this
// translation unit (the library is built with -ffp-contract=off for the bit-exact kNN / parity-critical inference kernels)
#pragma clang fp contract(fast)
#include "pps_fka_common.h"
#include "../../include/ppsurf_amd.h"

#define FT_NT 256
#define FT_TM 16
#define FT_GMAX_CAP 512                  // upper bound of blocks per shape (grid = G x B), each loops over its tiles

namespace {

__device__ __forceinline__ float act_grad(float y, int act) {
    if (act == 2) {
        const float s = 1.f / (1.f + __expf(-y));
        return s * (1.f + y * (1.f - s));
    }
    return y > 0.f ? 1.f : 0.f;
}

__device__ __forceinline__ float in_apply(float z, const float* st, int t, const float* geo, int wofs, int bofs, int K) {
    return K > 1 ? (z - st[2 * t]) * st[2 * t + 1] * geo[wofs + t] + geo[bofs + t] : z;
}

// [a ; max_j(a * dw)] for the 16 channels of this lane
__device__ __forceinline__ void pool16(const float (&a)[16], const Geo& g, float (&p)[16]) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float v = row16_max(g.valid ? a[t] * g.dw : -INFINITY);
        p[t] = v == -INFINITY ? 0.f : v;               // rows past the end of the shape: keep everything finite (0 * inf = NaN)
    }
}

// o[c] = sum_t w[t][c] * d[t]   (transposed 16x32 product)
__device__ __forceinline__ void fc32_t(const float (&d)[16], const float* w, float (&o)[32]) {
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) s += w[t * 32 + c] * d[t];
        o[c] = s;
    }
}

// first lane of the DPP row (lowest neighbour index) for which `hit` holds
__device__ __forceinline__ bool first_in_row(bool hit) {
    const unsigned long long bal = __ballot(hit);
    const int lane = threadIdx.x & 63;
    const unsigned row = (unsigned)(bal >> (lane & 48)) & 0xffffu;
    return hit && ((row & ((1u << (lane & 15)) - 1u)) == 0u);
}

// per-wave LDS staging area for the outer-product accumulation
struct Stage {
    float d[64][17];
    float i[64][33];
};

// acc[h][r] += sum over the wave's 64 lanes of dz[t] * in[c],  t = 4*(lane>>4)+r,  c = (lane&15) + 16*h   (NI = 32 or 3 inputs)
template <int NI>
__device__ __forceinline__ void outer_acc(const float (&dz)[16], const float* in, Stage* sg, f32x4 (&acc)[2]) {
    const int lane = threadIdx.x & 63;
    Stage& s = sg[threadIdx.x >> 6];
    __syncthreads();                                  // previous tile's fragment reads are done
#pragma unroll
    for (int t = 0; t < 16; ++t) s.d[lane][t] = dz[t];
#pragma unroll
    for (int c = 0; c < NI; ++c) s.i[lane][c] = in[c];
    __syncthreads();
    const int n = lane & 15, k = lane >> 4;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        const float a = s.d[4 * kb + k][n];
        if (NI == 32) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, s.i[4 * kb + k][n], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, s.i[4 * kb + k][16 + n], acc[1], 0, 0, 0);
        } else {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, n < NI ? s.i[4 * kb + k][n] : 0.f, acc[0], 0, 0, 0);
        }
    }
}

// sum the 4 waves' accumulators and write the block partial: out[t * ld + c]
template <int NI>
__device__ __forceinline__ void outer_store(const f32x4 (&acc)[2], float* red /* LDS [4][512] */, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * 512 + (4 * (lane >> 4) + r) * 32 + (lane & 15) + 16 * h] = acc[h][r];
    __syncthreads();
    for (int e = threadIdx.x; e < 512; e += FT_NT) {
        const int t = e >> 5, c = e & 31;
        if (c < NI) out[t * NI + c] = red[e] + red[512 + e] + red[1024 + e] + red[1536 + e];
    }
}

// block sums of 32 per-lane float accumulators (in double) -> part[32]
__device__ __forceinline__ void block_sum32(const float (&s1)[16], const float (&s2)[16], double* __restrict__ part, double* red /* LDS [4][32] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        double a = (double)s1[t], b = (double)s2[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (lane == 0) { red[wave * 32 + 2 * t] = a; red[wave * 32 + 2 * t + 1] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 32) part[threadIdx.x] = red[threadIdx.x] + red[32 + threadIdx.x] + red[64 + threadIdx.x] + red[96 + threadIdx.x];
}

__device__ __forceinline__ double block_sum1(float v, double* red /* LDS [4] */) {
    double a = (double)v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

struct Tile {
    int64_t mg;      // global support-point row (b*M + m)
    int64_t lim;     // (b+1)*M
    int j;
};

__device__ __forceinline__ Tile tile_of(int tile, int64_t M) {
    Tile t;
    const int64_t b = blockIdx.y;
    t.mg = b * M + (int64_t)tile * FT_TM + (threadIdx.x >> 4);
    t.lim = (b + 1) * M;
    t.j = threadIdx.x & 15;
    return t;
}

// ---- forward ---------------------------------------------------------------------------------------------------------

// sum over the support points of max_j |p_j - s|  (nn.py:605-609)
__global__ __launch_bounds__(FT_NT) void fka_radius_kernel(const float* __restrict__ pts, const float* __restrict__ sup,
                                                           const int64_t* __restrict__ idx, int64_t M, int K, double* __restrict__ part) {
    __shared__ double red[4];
    const int ntiles = (int)((M + FT_TM - 1) / FT_TM);
    float acc = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const Tile t = tile_of(tile, M);
        float d = 0.f;
        if (t.mg < t.lim && t.j < K) {
            const int64_t i = idx[t.mg * K + t.j];
            const float px = pts[i * 3] - sup[t.mg * 3], py = pts[i * 3 + 1] - sup[t.mg * 3 + 1], pz = pts[i * 3 + 2] - sup[t.mg * 3 + 2];
            d = sqrtf(px * px + py * py + pz * pz);
        }
        d = row16_max(d);
        if (t.j == 0 && t.mg < t.lim) acc += d;
    }
    const double s = block_sum1(acc, red);
    if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = s;
}

__global__ __launch_bounds__(64) void fka_fin_radius_kernel(const double* __restrict__ part, int n, double count, float momentum,
                                                            float* __restrict__ geo_w) {
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += part[i];
        geo_w[GEO_RADIUS] = geo_w[GEO_RADIUS] * (1.f - momentum) + (float)(s / count) * momentum;
    }
}

// PHASE 1: statistics of fc1 output; PHASE 2: statistics of fc2 output; PHASE 3: g = act(fc3) * dw -> out
template <int PHASE>
__global__ __launch_bounds__(FT_NT) void fka_fwd_kernel(const float* __restrict__ pts, const float* __restrict__ sup,
                                                        const int64_t* __restrict__ idx, int64_t M, int K, const float* __restrict__ geo_g,
                                                        const float* __restrict__ stat1, const float* __restrict__ stat2,
                                                        double* __restrict__ part, float* __restrict__ gout) {
    __shared__ double red[128];
    const float* st1 = stat1 + blockIdx.y * 32;
    const float* st2 = stat2 + blockIdx.y * 32;
    const int act = (int)geo_g[GEO_ACT];
    const int ntiles = (int)((M + FT_TM - 1) / FT_TM);
    float s1[16], s2[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) s1[t] = s2[t] = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // the 1140 parameters do not fit the SGPR file; the compiler hoists their scalar loads out of the tile loop and parks them
        // in VGPR lanes (v_readlane per use).  Forcing a reload per tile instead was measured 1.9x SLOWER (exposed s_load latency
        // at 2 waves/SIMD), and a copy in LDS 2x slower (the ds_reads get hoisted too: 256 VGPRs + 1.5 KB/lane of scratch), so the
        // hoisting is left alone.
        const float* geo = geo_g;
        const Tile tl = tile_of(tile, M);
        const Geo g = geometry(pts, sup, idx, tl.mg, tl.lim, tl.j, K, geo);
        float v[16];
        fc1_raw(g, geo, v);
        if (PHASE >= 2) {
            float p[16], o[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] = act_fn(in_apply(v[t], st1, t, geo, GEO_IN1W, GEO_IN1B, K), act);
            pool16(v, g, p);
            fc32(v, p, geo + GEO_FC2, o);
            if (PHASE == 3) {
#pragma unroll
                for (int t = 0; t < 16; ++t) o[t] = act_fn(in_apply(o[t], st2, t, geo, GEO_IN2W, GEO_IN2B, K), act);
                pool16(o, g, p);
                fc32(o, p, geo + GEO_FC3, v);
                if (g.valid) {
                    f32x4* dst = (f32x4*)(gout + (tl.mg * K + tl.j) * 16);
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4)
                        dst[t4] = f32x4{act_fn(v[4 * t4], act) * g.dw, act_fn(v[4 * t4 + 1], act) * g.dw, act_fn(v[4 * t4 + 2], act) * g.dw,
                                        act_fn(v[4 * t4 + 3], act) * g.dw};
                }
                continue;
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] = o[t];
        }
        if (g.valid) {
#pragma unroll
            for (int t = 0; t < 16; ++t) { s1[t] += v[t]; s2[t] += v[t] * v[t]; }
        }
    }
    if (PHASE < 3) block_sum32(s1, s2, part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 32, red);
}

// part [B][G][32] (sum, sum of squares | sum dy, sum dy*xhat interleaved per channel) -> out [B][32]
// MODE 0: (mean, rstd) with biased variance, eps 1e-5;  MODE 1: (sum/count, sum2/count)
template <int MODE>
__global__ __launch_bounds__(256) void fka_fin_stat_kernel(const double* __restrict__ part, int G, double count, float* __restrict__ out) {
    __shared__ double sub[8][32];
    const int s = threadIdx.x & 31, c = threadIdx.x >> 5;
    const double* p = part + (int64_t)blockIdx.x * G * 32;
    double acc = 0.0;
    for (int b = c; b < G; b += 8) acc += p[(int64_t)b * 32 + s];
    sub[c][s] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        double a = 0.0, q = 0.0;
        for (int i = 0; i < 8; ++i) { a += sub[i][2 * threadIdx.x]; q += sub[i][2 * threadIdx.x + 1]; }
        float* o = out + blockIdx.x * 32;
        if (MODE == 0) {
            const double mean = a / count;
            double var = q / count - mean * mean;
            if (var < 0.0) var = 0.0;
            o[2 * threadIdx.x] = (float)mean;
            o[2 * threadIdx.x + 1] = (float)(1.0 / sqrt(var + 1e-5));
        } else {
            o[2 * threadIdx.x] = (float)(a / count);
            o[2 * threadIdx.x + 1] = (float)(q / count);
        }
    }
}

// ---- backward --------------------------------------------------------------------------------------------------------

template <int PASS>
__global__ __launch_bounds__(FT_NT) void fka_bwd_kernel(const float* __restrict__ pts, const float* __restrict__ sup,
                                                        const int64_t* __restrict__ idx, int64_t M, int K, const float* __restrict__ geo_g,
                                                        const float* __restrict__ stat1, const float* __restrict__ stat2,
                                                        const float* __restrict__ gmean /* [B][32] of the IN being left, PASS 2/3 */,
                                                        const float* __restrict__ dg, float* __restrict__ dyb, float* __restrict__ ddwb,
                                                        double* __restrict__ part_s, float* __restrict__ part_w, double* __restrict__ part_ab) {
    __shared__ Stage stage[4];
    __shared__ float redw[4 * 512];
    __shared__ double red[128];
    const float* st1 = stat1 + blockIdx.y * 32;
    const float* st2 = stat2 + blockIdx.y * 32;
    const float* gm = gmean ? gmean + blockIdx.y * 32 : nullptr;
    const int act = (int)geo_g[GEO_ACT];
    const int ntiles = (int)((M + FT_TM - 1) / FT_TM);
    const int64_t blk = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    float s1[16], s2[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) s1[t] = s2[t] = 0.f;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float dalpha = 0.f, dbeta = 0.f;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const float* geo = geo_g;                                   // see fka_fwd_kernel
        const Tile tl = tile_of(tile, M);
        const Geo g = geometry(pts, sup, idx, tl.mg, tl.lim, tl.j, K, geo);
        const int64_t e = tl.mg * K + tl.j;                         // entry number (valid lanes only)
        float z1[16];
        fc1_raw(g, geo, z1);

        if (PASS == 3) {
            float dz1[16];
            if (g.valid) {
                const f32x4* src = (const f32x4*)(dyb + e * 16);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) { const f32x4 v = src[t4]; dz1[4 * t4] = v.x; dz1[4 * t4 + 1] = v.y; dz1[4 * t4 + 2] = v.z; dz1[4 * t4 + 3] = v.w; }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) dz1[t] = 0.f;
            }
            if (K > 1) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float xh = (z1[t] - st1[2 * t]) * st1[2 * t + 1];
                    dz1[t] = g.valid ? geo[GEO_IN1W + t] * st1[2 * t + 1] * (dz1[t] - gm[2 * t] - xh * gm[2 * t + 1]) : 0.f;
                }
            }
            outer_acc<3>(dz1, g.pn, stage, acc);
            // distance weights: dw_j = K u_j / s, u = sigmoid(-alpha d + beta)   (nn.py:619-624)
            float d = 0.f;
            if (g.valid) {
                const int64_t i = idx[e];
                const float px = pts[i * 3] - sup[tl.mg * 3], py = pts[i * 3 + 1] - sup[tl.mg * 3 + 1], pz = pts[i * 3 + 2] - sup[tl.mg * 3 + 2];
                d = sqrtf(px * px + py * py + pz * pz);
            }
            const float u = g.valid ? 1.f / (1.f + __expf(-(-geo[GEO_ALPHA] * d + geo[GEO_BETA]))) : 0.f;
            float ssum = row16_sum(u);
            ssum = ssum + (ssum == 0.f ? 1.f : 0.f) + 1e-6f;
            const float ddw = g.valid ? ddwb[e] : 0.f;
            const float tot = row16_sum(ddw * g.dw);
            const float du = ((float)K * ddw - tot) / ssum;
            const float da = du * u * (1.f - u);
            if (g.valid) { dalpha += -d * da; dbeta += da; }
            continue;
        }

        // PASS 1 and 2 share the first stage of the recomputation
        float a1[16], p1[16], y1[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) { y1[t] = in_apply(z1[t], st1, t, geo, GEO_IN1W, GEO_IN1B, K); a1[t] = act_fn(y1[t], act); }
        pool16(a1, g, p1);
        float z2[16];
        fc32(a1, p1, geo + GEO_FC2, z2);

        if (PASS == 1) {
            float a2[16], p2[16], y2[16], z3[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) { y2[t] = in_apply(z2[t], st2, t, geo, GEO_IN2W, GEO_IN2B, K); a2[t] = act_fn(y2[t], act); }
            pool16(a2, g, p2);
            fc32(a2, p2, geo + GEO_FC3, z3);
            float dz3[16], ddw = 0.f;
            if (g.valid) {
                const f32x4* src = (const f32x4*)(dg + e * 16);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) { const f32x4 v = src[t4]; dz3[4 * t4] = v.x; dz3[4 * t4 + 1] = v.y; dz3[4 * t4 + 2] = v.z; dz3[4 * t4 + 3] = v.w; }
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    ddw += dz3[t] * act_fn(z3[t], act);
                    dz3[t] = dz3[t] * g.dw * act_grad(z3[t], act);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) dz3[t] = 0.f;
            }
            float in3[32], din[32];
#pragma unroll
            for (int t = 0; t < 16; ++t) { in3[t] = a2[t]; in3[16 + t] = p2[t]; }
            outer_acc<32>(dz3, in3, stage, acc);
            fc32_t(dz3, geo + GEO_FC3, din);
            float dy2[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float dp = row16_sum(din[16 + t]);
                const bool first = first_in_row(g.valid && a2[t] * g.dw == p2[t]);
                const float da = din[t] + (first ? dp * g.dw : 0.f);
                ddw += first ? dp * a2[t] : 0.f;
                dy2[t] = g.valid ? da * act_grad(y2[t], act) : 0.f;
                if (K > 1 && g.valid) { s1[t] += dy2[t]; s2[t] += dy2[t] * (z2[t] - st2[2 * t]) * st2[2 * t + 1]; }
            }
            if (g.valid) {
                f32x4* dst = (f32x4*)(dyb + e * 16);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) dst[t4] = f32x4{dy2[4 * t4], dy2[4 * t4 + 1], dy2[4 * t4 + 2], dy2[4 * t4 + 3]};
                ddwb[e] = ddw;
            }
        } else {                                                     // PASS 2
            float dz2[16];
            if (g.valid) {
                const f32x4* src = (const f32x4*)(dyb + e * 16);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) { const f32x4 v = src[t4]; dz2[4 * t4] = v.x; dz2[4 * t4 + 1] = v.y; dz2[4 * t4 + 2] = v.z; dz2[4 * t4 + 3] = v.w; }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) dz2[t] = 0.f;
            }
            if (K > 1) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float xh = (z2[t] - st2[2 * t]) * st2[2 * t + 1];
                    dz2[t] = g.valid ? geo[GEO_IN2W + t] * st2[2 * t + 1] * (dz2[t] - gm[2 * t] - xh * gm[2 * t + 1]) : 0.f;
                }
            }
            float in2[32], din[32];
#pragma unroll
            for (int t = 0; t < 16; ++t) { in2[t] = a1[t]; in2[16 + t] = p1[t]; }
            outer_acc<32>(dz2, in2, stage, acc);
            fc32_t(dz2, geo + GEO_FC2, din);
            float dy1[16], ddw = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float dp = row16_sum(din[16 + t]);
                const bool first = first_in_row(g.valid && a1[t] * g.dw == p1[t]);
                const float da = din[t] + (first ? dp * g.dw : 0.f);
                ddw += first ? dp * a1[t] : 0.f;
                dy1[t] = g.valid ? da * act_grad(y1[t], act) : 0.f;
                if (K > 1 && g.valid) { s1[t] += dy1[t]; s2[t] += dy1[t] * (z1[t] - st1[2 * t]) * st1[2 * t + 1]; }
            }
            if (g.valid) {
                f32x4* dst = (f32x4*)(dyb + e * 16);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) dst[t4] = f32x4{dy1[4 * t4], dy1[4 * t4 + 1], dy1[4 * t4 + 2], dy1[4 * t4 + 3]};
                ddwb[e] += ddw;
            }
        }
    }
    if (PASS == 3) {
        outer_store<3>(acc, redw, part_w + blk * 48);
        const double a = block_sum1(dalpha, red);
        const double b = block_sum1(dbeta, red + 8);
        if (threadIdx.x == 0) { part_ab[blk * 2] = a; part_ab[blk * 2 + 1] = b; }
    } else {
        outer_store<32>(acc, redw, part_w + blk * 512);
        block_sum32(s1, s2, part_s + blk * 32, red);
    }
}

// dgeo[1140] from the per-block partials and the InstanceNorm sums: one wave per entry, lanes stride over the blocks and
// the 64 sub-sums are combined by a fixed butterfly (deterministic)
__global__ __launch_bounds__(256) void fka_fin_grad_kernel(const float* __restrict__ pw3, const float* __restrict__ pw2, const float* __restrict__ pw1,
                                                           const double* __restrict__ pab, int nblk, const float* __restrict__ gm2,
                                                           const float* __restrict__ gm1, int B, double count, int K, float* __restrict__ dgeo) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= GEO_FLOATS) return;
    double s = 0.0;
    if (e >= GEO_FC3 && e < GEO_FC3 + 512) {
        for (int b = lane; b < nblk; b += 64) s += (double)pw3[(int64_t)b * 512 + (e - GEO_FC3)];
    } else if (e >= GEO_FC2 && e < GEO_FC2 + 512) {
        for (int b = lane; b < nblk; b += 64) s += (double)pw2[(int64_t)b * 512 + (e - GEO_FC2)];
    } else if (e >= GEO_FC1 && e < GEO_FC1 + 48) {
        for (int b = lane; b < nblk; b += 64) s += (double)pw1[(int64_t)b * 48 + (e - GEO_FC1)];
    } else if (e == GEO_ALPHA || e == GEO_BETA) {
        for (int b = lane; b < nblk; b += 64) s += pab[(int64_t)b * 2 + (e == GEO_BETA ? 1 : 0)];
    } else if (e >= GEO_IN1W && K > 1) {                       // affine parameters of the InstanceNorms: sum over the shapes
        const int t = (e - GEO_IN1W) & 15, which = (e - GEO_IN1W) >> 4;   // 0: IN1 weight, 1: IN1 bias, 2: IN2 weight, 3: IN2 bias
        const float* gm = which < 2 ? gm1 : gm2;
        for (int b = lane; b < B; b += 64) s += (double)gm[b * 32 + 2 * t + ((which & 1) ? 0 : 1)] * count;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) dgeo[e] = (float)s;
}

// blocks per shape: the heavy kernels keep 2 workgroups per CU resident (225 VGPRs), so b * G is aimed at one full wave of
// 2 * CUs workgroups, each looping over its share of the tiles (960 workgroups in two uneven rounds were 10 % slower)
inline int grid_g(int64_t M, int64_t B) {
    const int64_t ntiles = (M + FT_TM - 1) / FT_TM;
    int cus = pps_device_cu_count();
    if (cus <= 0) cus = 256;
    int64_t g = (2 * (int64_t)cus) / (B > 0 ? B : 1);
    if (g < 1) g = 1;
    if (g > FT_GMAX_CAP) g = FT_GMAX_CAP;
    return (int)(ntiles < g ? ntiles : g);
}

inline char* align256(char* p) { return (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255); }

}  // namespace

extern "C" {

size_t pps_fka_train_ws_bytes(int64_t b, int64_t m, int k) {
    if (b < 1 || m < 1 || k < 1) return 0;
    const size_t nblk = (size_t)b * grid_g(m, b);
    // stat partials (double [nblk][32]) x2, radius partials, weight partials (512 + 512 + 48 floats), alpha/beta partials,
    // gradient means [b][32] x2, dy scratch [b*m*k][16], ddw scratch [b*m*k]
    return 4096 + nblk * (32 * 8 * 2 + 8 + (512 + 512 + 48) * 4 + 16) + (size_t)b * 32 * 4 * 2 + (size_t)b * m * k * 17 * 4;
}

int pps_fka_geometry_fwd_f32(const float* pts, const float* sup, const int64_t* idx, int64_t b, int64_t m, int k, float* geo_w,
                             float momentum, float* g_out, float* stat, void* ws, void* stream) {
    if (b < 0 || m < 0 || k < 1 || k > 16) return PPS_ERR_ARG;
    if (b == 0 || m == 0) return PPS_OK;
    if (!pts || !sup || !idx || !geo_w || !g_out || !stat || !ws) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int G = grid_g(m, b);
    const dim3 grid(G, (unsigned)b);
    double* part = (double*)align256((char*)ws);
    float* stat1 = stat;
    float* stat2 = stat + b * 32;
    const double count = (double)m * k;
    if (momentum > 0.f) {
        hipLaunchKernelGGL(fka_radius_kernel, grid, dim3(FT_NT), 0, st, pts, sup, idx, m, k, part);
        hipLaunchKernelGGL(fka_fin_radius_kernel, dim3(1), dim3(64), 0, st, (const double*)part, (int)(G * b), (double)(b * m), momentum, geo_w);
    }
    hipLaunchKernelGGL(fka_fwd_kernel<1>, grid, dim3(FT_NT), 0, st, pts, sup, idx, m, k, (const float*)geo_w, (const float*)stat1,
                       (const float*)stat2, part, (float*)nullptr);
    hipLaunchKernelGGL(fka_fin_stat_kernel<0>, dim3((unsigned)b), dim3(256), 0, st, (const double*)part, G, count, stat1);
    hipLaunchKernelGGL(fka_fwd_kernel<2>, grid, dim3(FT_NT), 0, st, pts, sup, idx, m, k, (const float*)geo_w, (const float*)stat1,
                       (const float*)stat2, part, (float*)nullptr);
    hipLaunchKernelGGL(fka_fin_stat_kernel<0>, dim3((unsigned)b), dim3(256), 0, st, (const double*)part, G, count, stat2);
    hipLaunchKernelGGL(fka_fwd_kernel<3>, grid, dim3(FT_NT), 0, st, pts, sup, idx, m, k, (const float*)geo_w, (const float*)stat1,
                       (const float*)stat2, part, g_out);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_fka_geometry_bwd_f32(const float* pts, const float* sup, const int64_t* idx, int64_t b, int64_t m, int k, const float* geo_w,
                             const float* stat, const float* dg, float* dgeo, void* ws, void* stream) {
    if (b < 0 || m < 0 || k < 1 || k > 16 || !dgeo) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (b == 0 || m == 0) return hipMemsetAsync(dgeo, 0, GEO_FLOATS * sizeof(float), st) == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
    if (!pts || !sup || !idx || !geo_w || !stat || !dg || !ws) return PPS_ERR_ARG;
    const int G = grid_g(m, b);
    const dim3 grid(G, (unsigned)b);
    const size_t nblk = (size_t)G * b;
    char* p = align256((char*)ws);
    double* part_s = (double*)p;            p += nblk * 32 * 8;
    double* part_ab = (double*)p;           p += nblk * 16;
    float* pw3 = (float*)p;                 p += nblk * 512 * 4;
    float* pw2 = (float*)p;                 p += nblk * 512 * 4;
    float* pw1 = (float*)p;                 p += nblk * 48 * 4;
    float* gm2 = (float*)p;                 p += (size_t)b * 32 * 4;
    float* gm1 = (float*)p;                 p += (size_t)b * 32 * 4;
    p = align256(p);
    float* dyb = (float*)p;                 p += (size_t)b * m * k * 16 * 4;
    float* ddwb = (float*)p;
    const float* stat1 = stat;
    const float* stat2 = stat + b * 32;
    const double count = (double)m * k;
    hipLaunchKernelGGL(fka_bwd_kernel<1>, grid, dim3(FT_NT), 0, st, pts, sup, idx, m, k, geo_w, stat1, stat2, (const float*)nullptr, dg, dyb, ddwb,
                       part_s, pw3, part_ab);
    hipLaunchKernelGGL(fka_fin_stat_kernel<1>, dim3((unsigned)b), dim3(256), 0, st, (const double*)part_s, G, count, gm2);
    hipLaunchKernelGGL(fka_bwd_kernel<2>, grid, dim3(FT_NT), 0, st, pts, sup, idx, m, k, geo_w, stat1, stat2, (const float*)gm2, dg, dyb, ddwb,
                       part_s, pw2, part_ab);
    hipLaunchKernelGGL(fka_fin_stat_kernel<1>, dim3((unsigned)b), dim3(256), 0, st, (const double*)part_s, G, count, gm1);
    hipLaunchKernelGGL(fka_bwd_kernel<3>, grid, dim3(FT_NT), 0, st, pts, sup, idx, m, k, geo_w, stat1, stat2, (const float*)gm1, dg, dyb, ddwb,
                       part_s, pw1, part_ab);
    hipLaunchKernelGGL(fka_fin_grad_kernel, dim3((GEO_FLOATS + 3) / 4), dim3(256), 0, st, (const float*)pw3, (const float*)pw2,
                       (const float*)pw1, (const double*)part_ab, (int)nblk, (const float*)gm2, (const float*)gm1, (int)b, count, k, dgeo);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

}  // extern "C"
