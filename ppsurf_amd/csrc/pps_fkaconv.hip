// FKAConv point-convolution encoder kernels for gfx950 (eval mode), point-major activations.
//
// replaces: source/base/nn.py:592-652 `FKAConvLayer.forward` (~25 ATen launches + two InstanceNorms per layer),
//           :655-697 batch_gather / max_pool / interpolate, and the 1x1 Conv1d + BatchNorm1d + ReLU + residual
//           glue of ResidualBlock / FKAConvNetwork (:438-450, :508-554).
//
// One FKAConv layer = 3 launches (the two InstanceNorm2d are GLOBAL reductions over all (support point, neighbour)
// pairs, nn.py:586-587,630,638):
//   phase 1: geometry -> fc1                      -> per-block partial sums of IN1 statistics
//   phase 2: ... IN1, act, max-pool over K, fc2   -> per-block partial sums of IN2 statistics
//   phase 3: ... IN2, act, max-pool, fc3 * dw     -> m3[m][j][16] in LDS, F = sum_j x[idx] (x) m3, out = Wcv . F
// Geometry and the tiny MLPs are recomputed in every phase instead of being stored (HBM traffic stays at the
// compulsory x/pts/idx reads).  Statistics are reduced in double, in a fixed order (deterministic).
// 16 lanes (one DPP row) handle the K <= 16 neighbours of one support point.
#include "pps_fka_common.h"
#include "../../include/ppsurf_amd.h"

// block partial sums (double) of v[t], v[t]^2 over valid lanes -> part[blockIdx.x][16][2]
__device__ __forceinline__ void block_stats(const float (&v)[16], bool valid, double* __restrict__ part, double* red /* LDS [4][32] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        double s = valid ? (double)v[t] : 0.0, q = valid ? (double)v[t] * (double)v[t] : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
        if (lane == 0) { red[wave * 32 + 2 * t] = s; red[wave * 32 + 2 * t + 1] = q; }
    }
    __syncthreads();
    if (threadIdx.x < 32)
        part[(int64_t)blockIdx.x * 32 + threadIdx.x] = red[threadIdx.x] + red[32 + threadIdx.x] + red[64 + threadIdx.x] + red[96 + threadIdx.x];
}

// mean / rstd (biased variance, eps 1e-5) of the 16 channels from the per-block partials: ONE workgroup, fixed summation order
// (thread t sums the partials of statistic t%32 over the blocks b = t/32 (mod 8), then the 8 sub-sums are added in order).
__global__ __launch_bounds__(256) void fka_finalize_kernel(const double* __restrict__ part, int nblk, double count, float* __restrict__ stat) {
    __shared__ double sub[8][32];
    const int s = threadIdx.x & 31, c = threadIdx.x >> 5;
    double acc = 0.0;
    for (int b = c; b < nblk; b += 8) acc += part[(int64_t)b * 32 + s];
    sub[c][s] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        double sm = 0.0, sq = 0.0;
        for (int i = 0; i < 8; ++i) { sm += sub[i][2 * threadIdx.x]; sq += sub[i][2 * threadIdx.x + 1]; }
        const double mean = sm / count;
        double var = sq / count - mean * mean;
        if (var < 0.0) var = 0.0;
        stat[2 * threadIdx.x] = (float)mean;
        stat[2 * threadIdx.x + 1] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

template <int PHASE>
__global__ __launch_bounds__(FK_NT) void fka_stats_kernel(const float* __restrict__ pts, const float* __restrict__ sup,
                                                          const int64_t* __restrict__ idx, int64_t M, int K,
                                                          const float* __restrict__ geo_g, const float* __restrict__ stat1,
                                                          double* __restrict__ part_out) {
    // the small per-layer parameters are read straight from the kernel-argument array with wave-uniform indices:
    // scalar loads (s_load) feeding SGPR operands instead of 1100 LDS broadcast reads per lane
    const float* __restrict__ geo = geo_g;
    __shared__ double red[128];
    const int j = threadIdx.x & 15;
    const int64_t m = (int64_t)blockIdx.x * FK_TM + (threadIdx.x >> 4);
    const Geo g = geometry(pts, sup, idx, m, M, j, K, geo);
    float v[16];
    fc1_raw(g, geo, v);
    if (PHASE == 2) {
        float mp[16], o[16];
        norm_act_pool(v, g, geo, stat1, GEO_IN1W, GEO_IN1B, K, mp);
        fc32(v, mp, geo + GEO_FC2, o);
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = o[t];
    }
    block_stats(v, g.valid, part_out, red);
}

// phase 3 + feature aggregation: F[m][c*16 + t] = sum_j x[idx[m][j]][c] * m3[m][j][t]   (nn.py:643-649)
// The (1,16) convolution that follows (nn.py:650) is the dense product F[M, Cin*16] x W^T, done by rows_gemm_kernel.
__global__ __launch_bounds__(FK_NT) void fka_feat_kernel(const float* __restrict__ x, const float* __restrict__ pts,
                                                         const float* __restrict__ sup, const int64_t* __restrict__ idx, int64_t M, int K,
                                                         int Cin, const float* __restrict__ geo_g, const float* __restrict__ stat1,
                                                         const float* __restrict__ stat2, float* __restrict__ F) {
    const float* __restrict__ geo = geo_g;
    __shared__ float m3[FK_TM][16][17];          // [m][j][t], padded
    __shared__ int nb[FK_TM][16];                // neighbour row (or -1)
    const int j = threadIdx.x & 15, ml = threadIdx.x >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * FK_TM;
    {
        const int64_t m = m0 + ml;
        const Geo g = geometry(pts, sup, idx, m, M, j, K, geo);
        float v[16], mp[16], o[16];
        fc1_raw(g, geo, v);
        norm_act_pool(v, g, geo, stat1, GEO_IN1W, GEO_IN1B, K, mp);
        fc32(v, mp, geo + GEO_FC2, o);
        norm_act_pool(o, g, geo, stat2, GEO_IN2W, GEO_IN2B, K, mp);
        fc32(o, mp, geo + GEO_FC3, v);
        const int act = (int)geo[GEO_ACT];
#pragma unroll
        for (int t = 0; t < 16; ++t) m3[ml][j][t] = g.valid ? act_fn(v[t], act) * g.dw : 0.f;     // nn.py:643
        nb[ml][j] = g.valid ? (int)idx[m * K + j] : -1;
    }
    __syncthreads();
    // thread = (support point mm, channel lane cl); 16 consecutive channels per pass -> coalesced x reads, 64 B F writes
    const int cl = threadIdx.x & 15, mm = threadIdx.x >> 4;
    if (m0 + mm >= M) return;
    float* frow = F + (m0 + mm) * (int64_t)Cin * 16;
    for (int c = cl; c < Cin; c += 16) {
        float f[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) f[t] = 0.f;
        for (int jj = 0; jj < K; ++jj) {
            const int r = nb[mm][jj];
            if (r >= 0) {
                const float xv = x[(int64_t)r * Cin + c];
#pragma unroll
                for (int t = 0; t < 16; ++t) f[t] += xv * m3[mm][jj][t];
            }
        }
        f32x4* dst = (f32x4*)(frow + (int64_t)c * 16);
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) dst[t4] = f32x4{f[4 * t4], f[4 * t4 + 1], f[4 * t4 + 2], f[4 * t4 + 3]};
    }
}

// out[m][o] = act(bias[o] + sum_k A[m][k] W[o][k] + residual[m][o]),  A[m] = [in1[idx1[m]] (c1) | in2[idx2[m]] (c2)], c1, c2 % 16 == 0.
// fp32 MFMA (v_mfma_f32_16x16x4_f32): one wave = 16 rows x (16*NOB) outputs, both operands straight from global/L2:
// A fragments from the packed weight image (pps_pack_dense_f32, 1 KiB contiguous per wave load), B fragments as one
// float4 of the (gathered) input row per lane.  grid = (row tiles of 64, output tiles of 16*NOB).
template <int NOB>
__global__ __launch_bounds__(256) void rows_gemm_kernel(const float* __restrict__ in1, const int64_t* __restrict__ idx1, int C1,
                                                        const float* __restrict__ in2, const int64_t* __restrict__ idx2, int C2,
                                                        const f32x4* __restrict__ wpack, const float* __restrict__ bias,
                                                        const float* __restrict__ residual, int act, int64_t M, int N,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * 16 + n;
    const int64_t rc = row < M ? row : M - 1;
    const int64_t r1 = idx1 ? idx1[rc] : rc;
    const int64_t r2 = in2 ? (idx2 ? idx2[rc] : rc) : 0;
    const int KB1 = C1 >> 4, KB = (C1 + C2) >> 4;
    const int ob0 = blockIdx.y * NOB;
    const f32x4* a1 = (const f32x4*)(in1 + r1 * C1) + g;
    const f32x4* a2 = in2 ? (const f32x4*)(in2 + r2 * C2) + g : a1;
    // FOUR independent accumulation chains per output block (k-steps 4kb, 4kb+1, 4kb+2, 4kb+3 of every 16-wide block), added
    // pairwise at the end: the contraction runs over up to K = 8192 (resnetb41.cv1), a single sequential fp32 chain of K/4 MFMAs
    // accumulates about twice the rounding error of the reference's blocked CPU GEMM (tools/enc_error_bisect.py); chains of K/16
    // halve it, and back-to-back MFMAs no longer wait on each other's result.
    f32x4 acc[NOB][4];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[ob][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* w = wpack + (int64_t)ob0 * KB * 64 + lane;
#pragma unroll 4
    for (int kb = 0; kb < KB; ++kb) {
        const f32x4 b = kb < KB1 ? a1[4 * kb] : a2[4 * (kb - KB1)];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const f32x4 a = w[((int64_t)ob * KB + kb) * 64];
            acc[ob][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[ob][0], 0, 0, 0);
            acc[ob][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[ob][1], 0, 0, 0);
            acc[ob][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[ob][2], 0, 0, 0);
            acc[ob][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[ob][3], 0, 0, 0);
        }
    }
    if (row >= M) return;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        const int o = 16 * (ob0 + ob) + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (o + r < N) {
                float v = (acc[ob][0][r] + acc[ob][1][r]) + (acc[ob][2][r] + acc[ob][3][r]);
                if (bias) v += bias[o + r];
                if (residual) v += residual[row * N + o + r];
                if (act == 1) v = fmaxf(v, 0.f);
                out[row * N + o + r] = v;
            }
        }
    }
}

// out[m][o] = act( bias[o] + sum_c A[m][c] * wt[c][o] + residual[m][o] ),  A = [in1[idx1[m]] | in2[idx2[m]]]
#define RL_TM 32
#define RL_TN 64
#define RL_TK 32
__global__ __launch_bounds__(256) void rows_linear_kernel(const float* __restrict__ in1, const int64_t* __restrict__ idx1, int C1,
                                                          const float* __restrict__ in2, const int64_t* __restrict__ idx2, int C2,
                                                          const float* __restrict__ wt, const float* __restrict__ bias,
                                                          const float* __restrict__ residual, int act, int64_t M, int Cout,
                                                          float* __restrict__ out) {
    __shared__ float As[RL_TM][RL_TK + 1];
    __shared__ float Ws[RL_TK][RL_TN + 1];
    __shared__ int64_t r1[RL_TM], r2[RL_TM];
    const int64_t m0 = (int64_t)blockIdx.x * RL_TM;
    const int o0 = blockIdx.y * RL_TN;
    if (threadIdx.x < RL_TM) {
        const int64_t m = m0 + threadIdx.x;
        const int64_t mc = m < M ? m : M - 1;
        r1[threadIdx.x] = idx1 ? idx1[mc] : mc;
        r2[threadIdx.x] = in2 ? (idx2 ? idx2[mc] : mc) : 0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // outputs 4*tx.., rows 2*ty..
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int Ct = C1 + C2;
    for (int k0 = 0; k0 < Ct; k0 += RL_TK) {
        for (int e = threadIdx.x; e < RL_TM * RL_TK; e += 256) {
            const int r = e / RL_TK, kk = e % RL_TK, c = k0 + kk;
            float v = 0.f;
            if (c < C1) v = in1[r1[r] * C1 + c];
            else if (c < Ct) v = in2[r2[r] * C2 + (c - C1)];
            As[r][kk] = v;
        }
        for (int e = threadIdx.x; e < RL_TK * RL_TN; e += 256) {
            const int kk = e / RL_TN, oo = e % RL_TN;
            Ws[kk][oo] = (k0 + kk < Ct && o0 + oo < Cout) ? wt[(int64_t)(k0 + kk) * Cout + o0 + oo] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < RL_TK; ++kk) {
            const float a0 = As[2 * ty][kk], a1 = As[2 * ty + 1][kk];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float w = Ws[kk][4 * tx + i];
                acc[0][i] += a0 * w;
                acc[1][i] += a1 * w;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int64_t m = m0 + 2 * ty + rr;
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = o0 + 4 * tx + i;
            if (o >= Cout) continue;
            float v = acc[rr][i] + (bias ? bias[o] : 0.f);
            if (residual) v += residual[m * Cout + o];
            if (act == 1) v = fmaxf(v, 0.f);
            out[m * Cout + o] = v;
        }
    }
}

// out[m][c] = max_j x[idx[m][j]][c]   (nn.py:677-680; K arbitrary; also the global max of nn.py:531 with one row of idx)
__global__ __launch_bounds__(256) void gather_max_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx, int64_t M, int K, int C,
                                                         float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * C) return;
    const int64_t m = e / C;
    const int c = (int)(e % C);
    float v = -INFINITY;
    for (int j = 0; j < K; ++j) v = fmaxf(v, x[idx[m * K + j] * C + c]);
    out[e] = v;
}

extern "C" {

size_t pps_fkaconv_geo_floats(void) { return GEO_FLOATS; }

size_t pps_fkaconv_ws_bytes(int64_t M, int cin) {
    const int64_t nblk = (M + FK_TM - 1) / FK_TM;
    return (size_t)nblk * 32 * sizeof(double) * 2 + 64 * sizeof(float) + (size_t)M * cin * 16 * sizeof(float);
}

static int launch_rows_gemm(const float* in1, const int64_t* idx1, int c1, const float* in2, const int64_t* idx2, int c2,
                            const float* wpack, const float* bias, const float* residual, int act, int64_t m, int n, float* out,
                            hipStream_t st) {
    const int obt = ((n + 31) / 32) * 2;                    // packed output blocks (out padded to 32)
    const unsigned gx = (unsigned)((m + 63) / 64);
    // few row tiles -> narrow output tiles so that the weight stream is spread over more CUs
    if (gx * (obt / 4) >= 256 && obt % 4 == 0)
        hipLaunchKernelGGL(rows_gemm_kernel<4>, dim3(gx, obt / 4), dim3(256), 0, st, in1, idx1, c1, in2, idx2, c2, (const f32x4*)wpack,
                           bias, residual, act, m, n, out);
    else
        hipLaunchKernelGGL(rows_gemm_kernel<2>, dim3(gx, obt / 2), dim3(256), 0, st, in1, idx1, c1, in2, idx2, c2, (const f32x4*)wpack,
                           bias, residual, act, m, n, out);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_fkaconv_fwd_f32(const float* x, const float* pts, const float* sup, const int64_t* idx, int64_t n, int64_t m, int k,
                        int cin, int cout, const float* geo, const float* wpack, const float* bias, int act_out, float* out,
                        void* ws, void* stream) {
    if (!x || !pts || !sup || !idx || !geo || !wpack || !out || !ws || n < 1 || m < 1 || k < 1 || k > 16 || cin < 1 || cout < 1)
        return PPS_ERR_ARG;
    const int nblk = (int)((m + FK_TM - 1) / FK_TM);
    double* part1 = (double*)ws;
    double* part2 = part1 + (size_t)nblk * 32;
    float* stat1 = (float*)(part2 + (size_t)nblk * 32);
    float* stat2 = stat1 + 32;
    float* F = stat2 + 32;
    hipStream_t st = (hipStream_t)stream;
    const double count = (double)m * k;
    hipLaunchKernelGGL(fka_stats_kernel<1>, dim3(nblk), dim3(FK_NT), 0, st, pts, sup, idx, m, k, geo, (const float*)nullptr, part1);
    hipLaunchKernelGGL(fka_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)part1, nblk, count, stat1);
    hipLaunchKernelGGL(fka_stats_kernel<2>, dim3(nblk), dim3(FK_NT), 0, st, pts, sup, idx, m, k, geo, (const float*)stat1, part2);
    hipLaunchKernelGGL(fka_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)part2, nblk, count, stat2);
    hipLaunchKernelGGL(fka_feat_kernel, dim3(nblk), dim3(FK_NT), 0, st, x, pts, sup, idx, m, k, cin, geo, (const float*)stat1,
                       (const float*)stat2, F);
    if (hipGetLastError() != hipSuccess) return PPS_ERR_LAUNCH;
    return launch_rows_gemm(F, nullptr, cin * 16, nullptr, nullptr, 0, wpack, bias, nullptr, act_out, m, cout, out, st);
}

int pps_rows_gemm_f32(const float* in1, const int64_t* idx1, int c1, const float* in2, const int64_t* idx2, int c2,
                      const float* wpack, const float* bias, const float* residual, int act, int64_t m, int cout, float* out,
                      void* stream) {
    if (!in1 || !wpack || !out || m < 1 || c1 < 16 || (c1 & 15) || c2 < 0 || (c2 & 15) || cout < 1 || (c2 > 0 && !in2)) return PPS_ERR_ARG;
    return launch_rows_gemm(in1, idx1, c1, c2 > 0 ? in2 : nullptr, idx2, c2, wpack, bias, residual, act, m, cout, out, (hipStream_t)stream);
}

int pps_rows_linear_f32(const float* in1, const int64_t* idx1, int c1, const float* in2, const int64_t* idx2, int c2,
                        const float* wt, const float* bias, const float* residual, int act, int64_t m, int cout, float* out,
                        void* stream) {
    if (!in1 || !wt || !out || m < 1 || c1 < 1 || c2 < 0 || cout < 1 || (c2 > 0 && !in2)) return PPS_ERR_ARG;
    dim3 grid((unsigned)((m + RL_TM - 1) / RL_TM), (unsigned)((cout + RL_TN - 1) / RL_TN));
    hipLaunchKernelGGL(rows_linear_kernel, grid, dim3(256), 0, (hipStream_t)stream, in1, idx1, c1, c2 > 0 ? in2 : nullptr, idx2, c2, wt,
                       bias, residual, act, m, cout, out);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_gather_max_f32(const float* x, const int64_t* idx, int64_t m, int k, int c, float* out, void* stream) {
    if (!x || !idx || !out || m < 1 || k < 1 || c < 1) return PPS_ERR_ARG;
    hipLaunchKernelGGL(gather_max_kernel, dim3((unsigned)((m * c + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, idx, m, k, c, out);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

}  // extern "C"
