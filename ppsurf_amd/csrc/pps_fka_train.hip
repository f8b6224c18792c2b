// FKAConv geometry branch for TRAINING: forward in train() mode and hand-written backward (gfx950).
//
// replaces (reference, under autograd): source/base/nn.py:601-643 -- neighbour offsets, norm_radius EMA (:608-613), distance
// weights, fc1 -> InstanceNorm -> act -> weighted max-pool -> fc2 -> InstanceNorm -> act -> max-pool -> fc3 -> act * dw,
// i.e. everything of FKAConvLayer.forward that produces the [M, K, 16] kernel-weighting matrix `g`; ~60 ATen launches
// forward and ~150 backward per layer on [B, M, K, 16] tensors (12 ms per 10 x 10k-point layer) become 4 + 3 launches.
//
// Mapping: a WAVE works on one support point at a time.  Lane (j, q) = (lane & 15, lane >> 4) holds, for neighbour j of the point, the four
// channels 4 q .. 4 q + 3 of every 16-channel activation -- which is both the C/D layout of v_mfma_f32_16x16x4_f32 for the matrix
// [channel][neighbour] and, with the contraction index ordered (step s, group q) <-> channel 4 q + s, its B operand layout.  So the two
// 32 -> 16 layers (and their transposes in the backward pass) are chains of fp32 MFMAs whose A operands, the layer's weights, sit in 8
// registers per lane each; pooling over the neighbours is a DPP row reduction of 4 values per lane.  (An earlier version kept a whole
// 16-channel row per lane and multiplied on the VALU: the 1140 parameters do not fit the SGPR file, the compiler parked them in VGPR lanes
// and the backward passes executed 4.6 v_readlane per FMA.)  fp32 MFMA is an exact fmaf chain, the arithmetic is unchanged.
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
#define FT_TM 4                          // support points per workgroup iteration: one per wave
#define FT_GMAX_CAP 512                  // upper bound of blocks per shape (grid = G x B), each loops over its tiles

namespace {

__device__ __forceinline__ float act_grad(float y, int act) {
    if (act == 2) {
        const float s = 1.f / (1.f + __expf(-y));
        return s * (1.f + y * (1.f - s));
    }
    return y > 0.f ? 1.f : 0.f;
}

// first lane of the DPP row (lowest neighbour index) for which `hit` holds
__device__ __forceinline__ bool first_in_row(bool hit) {
    const unsigned long long bal = __ballot(hit);
    const int lane = threadIdx.x & 63;
    const unsigned row = (unsigned)(bal >> (lane & 48)) & 0xffffu;
    return hit && ((row & ((1u << (lane & 15)) - 1u)) == 0u);
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
    t.mg = b * M + (int64_t)tile * 16 + (threadIdx.x >> 4);        // 16 points per workgroup iteration, one DPP row each (fka_radius_kernel)
    t.lim = (b + 1) * M;
    t.j = threadIdx.x & 15;
    return t;
}

// ---- forward ---------------------------------------------------------------------------------------------------------

// sum over the support points of max_j |p_j - s|  (nn.py:605-609)
__global__ __launch_bounds__(FT_NT) void fka_radius_kernel(const float* __restrict__ pts, const float* __restrict__ sup,
                                                           const int64_t* __restrict__ idx, int64_t M, int K, double* __restrict__ part) {
    __shared__ double red[4];
    const int ntiles = (int)((M + 15) / 16);
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
    double s = 0.0;                                    // lanes stride over the block partials, fixed butterfly: deterministic
    for (int i = threadIdx.x; i < n; i += 64) s += part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) geo_w[GEO_RADIUS] = geo_w[GEO_RADIUS] * (1.f - momentum) + (float)(s / count) * momentum;
}

// ---- one support point per wave -----------------------------------------------------------------------------------------

struct Pt {
    float dw;        // normalised distance weight of neighbour j (nn.py:619-624)
    float pn[3];     // neighbour offset / norm_radius (nn.py:601,616)
    float d, u, ssum;
    bool valid;
    int64_t e;       // entry number mg * K + j (valid lanes)
};

__device__ __forceinline__ Pt point_geometry(const float* __restrict__ pts, const float* __restrict__ sup, const int64_t* __restrict__ idx,
                                             int64_t mg, int64_t lim, int j, int K, float radius, float alpha, float beta) {
    Pt r;
    r.valid = (mg < lim) && (j < K);
    r.e = mg * K + j;
    r.d = 0.f;
    r.pn[0] = r.pn[1] = r.pn[2] = 0.f;
    if (r.valid) {
        const int64_t i = idx[r.e];
        const float px = pts[i * 3] - sup[mg * 3], py = pts[i * 3 + 1] - sup[mg * 3 + 1], pz = pts[i * 3 + 2] - sup[mg * 3 + 2];
        r.d = sqrtf(px * px + py * py + pz * pz);
        r.pn[0] = px / radius; r.pn[1] = py / radius; r.pn[2] = pz / radius;
    }
    r.u = r.valid ? 1.f / (1.f + __expf(-(-alpha * r.d + beta))) : 0.f;
    float sm = row16_sum(r.u);
    sm = sm + (sm == 0.f ? 1.f : 0.f) + 1e-6f;
    r.ssum = sm;
    r.dw = r.u / sm * (float)K;
    return r;
}

// the layer's parameters as this lane needs them (lane (j, q): channels t = 4 q + r; as MFMA row m = lane & 15)
struct LaneParams {
    float w1[4][3];              // fc1 rows of the lane's channels
    float in1w[4], in1b[4], in2w[4], in2b[4];
    float w2a[8], w3a[8];        // A operands of fc2 / fc3: W[m][in], in = (s < 4 ? 4 q + s : 16 + 4 q + s - 4) for contraction step s
    float w2t[2][4], w3t[2][4];  // A operands of the transposed products: W[4 q + s][16 mb + m]
    float radius, alpha, beta;
    int act;
};

template <bool BWD>
__device__ __forceinline__ LaneParams load_params(const float* __restrict__ geo) {
    LaneParams P;
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = 4 * q + r;
        P.w1[r][0] = geo[GEO_FC1 + t * 3]; P.w1[r][1] = geo[GEO_FC1 + t * 3 + 1]; P.w1[r][2] = geo[GEO_FC1 + t * 3 + 2];
        P.in1w[r] = geo[GEO_IN1W + t]; P.in1b[r] = geo[GEO_IN1B + t]; P.in2w[r] = geo[GEO_IN2W + t]; P.in2b[r] = geo[GEO_IN2B + t];
    }
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
        const int in = s2 < 4 ? 4 * q + s2 : 16 + 4 * q + (s2 - 4);
        P.w2a[s2] = geo[GEO_FC2 + m * 32 + in];
        P.w3a[s2] = geo[GEO_FC3 + m * 32 + in];
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            P.w2t[mb][s2] = BWD ? geo[GEO_FC2 + (4 * q + s2) * 32 + 16 * mb + m] : 0.f;
            P.w3t[mb][s2] = BWD ? geo[GEO_FC3 + (4 * q + s2) * 32 + 16 * mb + m] : 0.f;
        }
    P.radius = geo[GEO_RADIUS]; P.alpha = geo[GEO_ALPHA]; P.beta = geo[GEO_BETA];
    P.act = (int)geo[GEO_ACT];
    return P;
}

__device__ __forceinline__ f32x4 fc1_lane(const LaneParams& P, const Pt& g) {
    f32x4 z;
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = P.w1[r][0] * g.pn[0] + P.w1[r][1] * g.pn[1] + P.w1[r][2] * g.pn[2];
    return z;
}

// z[cout][j] = sum_in W[cout][in] * [a ; p][in][j]: 8 MFMAs, the activations are B operands as they are
__device__ __forceinline__ f32x4 fc32_mfma(const float (&wa)[8], const f32x4& a, const f32x4& p) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) z = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s2], a[s2], z, 0, 0, 0);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) z = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[4 + s2], p[s2], z, 0, 0, 0);
    return z;
}

// din[in][j] = sum_cout W[cout][in] * dz[cout][j] for the block mb of 16 inputs
__device__ __forceinline__ f32x4 fc32_t_mfma(const float (&wt)[4], const f32x4& dz) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) o = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[s2], dz[s2], o, 0, 0, 0);
    return o;
}

__device__ __forceinline__ f32x4 in_act(const f32x4& z, const float* st /* [16][2] mean, rstd of the shape */, const float (&w)[4], const float (&b)[4], int q,
                                       int K, int act, f32x4& y) {
    f32x4 a;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = 4 * q + r;
        y[r] = K > 1 ? (z[r] - st[2 * t]) * st[2 * t + 1] * w[r] + b[r] : z[r];
        a[r] = act_fn(y[r], act);
    }
    return a;
}

__device__ __forceinline__ f32x4 pool4(const f32x4& a, const Pt& g) {
    f32x4 p;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float v = row16_max(g.valid ? a[r] * g.dw : -INFINITY);
        p[r] = v == -INFINITY ? 0.f : v;               // points past the end of the shape: keep everything finite
    }
    return p;
}

// block sums of the 2 x 4 per-lane accumulators (channels 4 q + r) in double -> part[32] = (s1[t], s2[t]) interleaved per channel t
__device__ __forceinline__ void block_sum_lane4(const float (&s1)[4], const float (&s2)[4], double* __restrict__ part, double* red /* LDS [4][32] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double a = (double)s1[r], b = (double)s2[r];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if ((lane & 15) == 0) { red[wave * 32 + 2 * (4 * q + r)] = a; red[wave * 32 + 2 * (4 * q + r) + 1] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 32) part[threadIdx.x] = red[threadIdx.x] + red[32 + threadIdx.x] + red[64 + threadIdx.x] + red[96 + threadIdx.x];
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
    const LaneParams P = load_params<false>(geo_g);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    const int ntiles = (int)((M + FT_TM - 1) / FT_TM);
    const int64_t lim = ((int64_t)blockIdx.y + 1) * M;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t mg = (int64_t)blockIdx.y * M + (int64_t)tile * FT_TM + wave;
        const Pt g = point_geometry(pts, sup, idx, mg, lim, j, K, P.radius, P.alpha, P.beta);
        f32x4 v = fc1_lane(P, g);
        if (PHASE >= 2) {
            f32x4 y;
            const f32x4 a1 = in_act(v, st1, P.in1w, P.in1b, q, K, P.act, y);
            const f32x4 p1 = pool4(a1, g);
            v = fc32_mfma(P.w2a, a1, p1);
            if (PHASE == 3) {
                const f32x4 a2 = in_act(v, st2, P.in2w, P.in2b, q, K, P.act, y);
                const f32x4 p2 = pool4(a2, g);
                const f32x4 z3 = fc32_mfma(P.w3a, a2, p2);
                if (g.valid)
                    *(f32x4*)(gout + g.e * 16 + 4 * q) = f32x4{act_fn(z3[0], P.act) * g.dw, act_fn(z3[1], P.act) * g.dw, act_fn(z3[2], P.act) * g.dw,
                                                              act_fn(z3[3], P.act) * g.dw};
                continue;
            }
        }
        if (g.valid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[r] += v[r]; s2[r] += v[r] * v[r]; }
        }
    }
    if (PHASE < 3) block_sum_lane4(s1, s2, part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 32, red);
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

// per-wave staging area for the weight-gradient outer products: the matrices [channel][neighbour] the lanes hold in C/D layout are
// needed with the NEIGHBOUR as contraction index, i.e. transposed
struct Stage {
    float d[16][17];     // dz[cout][j]
    float i[32][17];     // in[k][j]
};

// acc[h][.] += sum_j dz[cout][j] * in[16 h + n][j]   (D rows = couts, columns = inputs;  NI = 32: both halves, NI = 3: pn, h = 0 only)
template <int NI>
__device__ __forceinline__ void outer_acc(const f32x4& dz, const f32x4& in_a, const f32x4& in_p, Stage& sg, f32x4 (&acc)[2]) {
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sg.d[4 * q + r][j] = dz[r];
        sg.i[4 * q + r][j] = in_a[r];                    // NI = 3: rows 0..2 = pn (the caller passes it in the q = 0 lanes), the rest 0
        if (NI == 32) sg.i[16 + 4 * q + r][j] = in_p[r];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {                     // contraction step: neighbours 4 s + q;  lane & 15 = cout (A) / input (B)
        const float a = sg.d[j][4 * s2 + q];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sg.i[j][4 * s2 + q], acc[0], 0, 0, 0);
        if (NI == 32) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sg.i[16 + j][4 * s2 + q], acc[1], 0, 0, 0);
    }
}

// sum over the four channel quarters of a per-neighbour value (lanes j, j + 16, j + 32, j + 48)
__device__ __forceinline__ float quarters_sum(float v) {            // row and half swaps of gfx950 instead of two ds_bpermute round trips
    const int lane = threadIdx.x & 63;
    v += __uint_as_float(pps::lane_xor_u32(__float_as_uint(v), 16, lane));
    v += __uint_as_float(pps::lane_xor_u32(__float_as_uint(v), 32, lane));
    return v;
}

// gradient of [a ; max_j(a * dw)] wrt a and dw: din_a + (this neighbour holds the maximum ? sum_j din_p * dw : 0)   (nn.py:631-633)
__device__ __forceinline__ f32x4 pool_grad(const f32x4& din_a, const f32x4& din_p, const f32x4& a, const f32x4& p, const Pt& g, float& ddw) {
    f32x4 da;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float dp = row16_sum(din_p[r]);
        const bool first = first_in_row(g.valid && a[r] * g.dw == p[r]);
        da[r] = din_a[r] + (first ? dp * g.dw : 0.f);
        ddw += first ? dp * a[r] : 0.f;
    }
    return da;
}

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
    const LaneParams P = load_params<true>(geo_g);
    const int act = P.act;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
    Stage& sg = stage[wave];
    const int ntiles = (int)((M + FT_TM - 1) / FT_TM);
    const int64_t lim = ((int64_t)blockIdx.y + 1) * M;
    const int64_t blk = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float dalpha = 0.f, dbeta = 0.f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t mg = (int64_t)blockIdx.y * M + (int64_t)tile * FT_TM + wave;
        const Pt g = point_geometry(pts, sup, idx, mg, lim, j, K, P.radius, P.alpha, P.beta);
        const f32x4 z1 = fc1_lane(P, g);

        if (PASS == 3) {
            f32x4 dz1 = g.valid ? *(const f32x4*)(dyb + g.e * 16 + 4 * q) : zero4;
            if (K > 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 4 * q + r;
                    const float xh = (z1[r] - st1[2 * t]) * st1[2 * t + 1];
                    dz1[r] = g.valid ? P.in1w[r] * st1[2 * t + 1] * (dz1[r] - gm[2 * t] - xh * gm[2 * t + 1]) : 0.f;
                }
            }
            const f32x4 pn4 = q == 0 ? f32x4{g.pn[0], g.pn[1], g.pn[2], 0.f} : zero4;      // input rows 0..2 of the staging area
            outer_acc<3>(dz1, pn4, zero4, sg, acc);
            // distance weights: dw_j = K u_j / s, u = sigmoid(-alpha d + beta)   (nn.py:619-624); the four quarters hold the same values
            const float ddw = g.valid ? ddwb[g.e] : 0.f;
            const float tot = row16_sum(ddw * g.dw);
            const float du = ((float)K * ddw - tot) / g.ssum;
            const float da = du * g.u * (1.f - g.u);
            if (g.valid && q == 0) { dalpha += -g.d * da; dbeta += da; }
            continue;
        }

        // PASS 1 and 2 share the first stage of the recomputation
        f32x4 y1;
        const f32x4 a1 = in_act(z1, st1, P.in1w, P.in1b, q, K, act, y1);
        const f32x4 p1 = pool4(a1, g);
        const f32x4 z2 = fc32_mfma(P.w2a, a1, p1);

        if (PASS == 1) {
            f32x4 y2;
            const f32x4 a2 = in_act(z2, st2, P.in2w, P.in2b, q, K, act, y2);
            const f32x4 p2 = pool4(a2, g);
            const f32x4 z3 = fc32_mfma(P.w3a, a2, p2);
            f32x4 dz3 = g.valid ? *(const f32x4*)(dg + g.e * 16 + 4 * q) : zero4;
            float ddw = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ddw += dz3[r] * act_fn(z3[r], act);
                dz3[r] = g.valid ? dz3[r] * g.dw * act_grad(z3[r], act) : 0.f;
            }
            outer_acc<32>(dz3, a2, p2, sg, acc);
            const f32x4 din_a = fc32_t_mfma(P.w3t[0], dz3), din_p = fc32_t_mfma(P.w3t[1], dz3);
            const f32x4 da = pool_grad(din_a, din_p, a2, p2, g, ddw);
            f32x4 dy2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 4 * q + r;
                dy2[r] = g.valid ? da[r] * act_grad(y2[r], act) : 0.f;
                if (K > 1 && g.valid) { s1[r] += dy2[r]; s2[r] += dy2[r] * (z2[r] - st2[2 * t]) * st2[2 * t + 1]; }
            }
            ddw = quarters_sum(ddw);
            if (g.valid) {
                *(f32x4*)(dyb + g.e * 16 + 4 * q) = dy2;
                if (q == 0) ddwb[g.e] = ddw;
            }
        } else {                                                     // PASS 2
            f32x4 dz2 = g.valid ? *(const f32x4*)(dyb + g.e * 16 + 4 * q) : zero4;
            if (K > 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 4 * q + r;
                    const float xh = (z2[r] - st2[2 * t]) * st2[2 * t + 1];
                    dz2[r] = g.valid ? P.in2w[r] * st2[2 * t + 1] * (dz2[r] - gm[2 * t] - xh * gm[2 * t + 1]) : 0.f;
                }
            }
            outer_acc<32>(dz2, a1, p1, sg, acc);
            const f32x4 din_a = fc32_t_mfma(P.w2t[0], dz2), din_p = fc32_t_mfma(P.w2t[1], dz2);
            float ddw = 0.f;
            const f32x4 da = pool_grad(din_a, din_p, a1, p1, g, ddw);
            f32x4 dy1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 4 * q + r;
                dy1[r] = g.valid ? da[r] * act_grad(y1[r], act) : 0.f;
                if (K > 1 && g.valid) { s1[r] += dy1[r]; s2[r] += dy1[r] * (z1[r] - st1[2 * t]) * st1[2 * t + 1]; }
            }
            ddw = quarters_sum(ddw);
            if (g.valid) {
                *(f32x4*)(dyb + g.e * 16 + 4 * q) = dy1;
                if (q == 0) ddwb[g.e] += ddw;
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
        block_sum_lane4(s1, s2, part_s + blk * 32, red);
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

// blocks per shape: b * G is aimed at ONE full wave of resident workgroups, each looping over its share of the tiles.  A wave carries one support
// point at a time through a serial chain (gather, sixteen dependent fp32 MFMAs, DPP reductions: ~1500 clocks), so the kernels run at the rate of
// the waves a SIMD holds; a workgroup is one wave per SIMD, and the register counts allow 5 (forward: 44 / 70 / 79 VGPRs) and 3 (backward: 132 /
// 120 / 61 VGPRs, 22 KB of LDS) workgroups per CU.  (Until round 5 the grid was 2 per CU for every kernel -- the limit of the 225-VGPR backward
// passes that preceded the MFMA form; more workgroups than are resident run in uneven rounds, which measured 10 % slower.)
#ifndef PPS_FKA_WG_FWD
#define PPS_FKA_WG_FWD 5
#define PPS_FKA_WG_BWD 3
#endif
constexpr int FT_WG_PER_CU_FWD = PPS_FKA_WG_FWD, FT_WG_PER_CU_BWD = PPS_FKA_WG_BWD;
inline int grid_g(int64_t M, int64_t B, int per_cu) {
    const int64_t ntiles = (M + FT_TM - 1) / FT_TM;
    int cus = pps_device_cu_count();
    if (cus <= 0) cus = 256;
    int64_t g = ((int64_t)per_cu * cus) / (B > 0 ? B : 1);
    if (g < 1) g = 1;
    if (g > FT_GMAX_CAP) g = FT_GMAX_CAP;
    return (int)(ntiles < g ? ntiles : g);
}

inline char* align256(char* p) { return (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255); }

}  // namespace

extern "C" {

size_t pps_fka_train_ws_bytes(int64_t b, int64_t m, int k) {
    if (b < 1 || m < 1 || k < 1) return 0;
    const size_t nblk = (size_t)b * grid_g(m, b, FT_WG_PER_CU_FWD > FT_WG_PER_CU_BWD ? FT_WG_PER_CU_FWD : FT_WG_PER_CU_BWD);
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
    const int G = grid_g(m, b, FT_WG_PER_CU_FWD);
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
    const int G = grid_g(m, b, FT_WG_PER_CU_BWD);
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
