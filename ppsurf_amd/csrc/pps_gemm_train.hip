// Dense layers of the training step for ANY layer shape: the three products of a linear layer y = x W^T + b over rows, in 16-bit storage
// (bfloat16 for trainer.precision bf16-mixed, IEEE half for 16-mixed) with fp32 accumulation on the matrix pipe.
//
// replaces: the library GEMMs (hipBLASLt `Cijk_*` kernels behind F.linear / torch.mm / torch.bmm) that ppsurf_amd/train_graph.py::_RowsLinear ran for
// the layers the fused row kernels of pps_rows_train.hip do not take -- every 1x1 Conv1d / Linear of the FKAConv encoder and its (1,16) Conv2d
// (source/base/nn.py:438-450, 508-554, 571, 650: 3*16 = 48 ... 8192 input channels, 32 ... 1024 output channels, 390 ... 100 000 rows per batch),
// the per-point table of the interpolation head, fc_value / fc8 (source/poco_model.py:405-417), the STN's fully connected layers
// (source/base/nn.py:183-188: 64 -> 4096), att.fc_value and the MLP (nn.py:376-417: 256 -> 2).
//
//   pps_gemm_nt_16   y [M, N] = x [M, K] w [N, K]^T (+ bias)        forward;  and the input gradient dx = g w with the TRANSPOSED image of w
//   pps_gemm_tn_16   dw [N, K] = g [M, N]^T x [M, K]   (fp32)       weight gradient: contraction over the rows
//   pps_transpose_cast_pieces   fp32 master weights [N, K] -> 16-bit images [K, N] of all layers in one launch (table of matrices)
//
// NT: both operands are K-contiguous, so MFMA fragments are read straight from global memory (16 bytes per lane: 8 consecutive k of one row /
// one output channel); a wave owns 16 TN rows x 16 TM channels, grid = (row tiles, channel blocks), the weights come from L2.  The waves of a
// workgroup are stacked over the rows for the wide layers and split the contraction for the coarse encoder levels (see gemm_nt_kernel).
// TN: the contraction index is the ROW, which neither operand has contiguous: 64-row tiles of g and x are staged in LDS row-major and fragments are
// taken with ds_read_b64_tr_b16 (the LDS transpose read of gfx950), as in rows_dw_kernel; a workgroup owns a 64 x 64 block of dw over a slab of
// rows, slab partials are summed in a fixed order (no atomics: bit-reproducible).
#include "pps_common.h"
#include "../../include/ppsurf_amd.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const f16x8_t*)&a, *(const f16x8_t*)&b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&a, *(const bf16x8_t*)&b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ unsigned pack2(float a, float b) {                     // round to nearest even
    const f32x2 v = {a, b};
    if constexpr (F16) { const f16x2_t h = __builtin_convertvector(v, f16x2_t); return *(const unsigned*)&h; }
    else { const bf16x2_t h = __builtin_convertvector(v, bf16x2_t); return *(const unsigned*)&h; }
}

// ---------------------------------------------------------------------------------------------------------------------
// NT product.  D[m <-> output channel][n <-> row]: A = w (lane (m = l & 15, kg = l >> 4): 8 consecutive k of one channel), B = x (lane (n, kg): 8
// consecutive k of row n).  A workgroup is 4 waves against one block of 16 TM channels:
//   KSPLIT = 1   the waves are stacked over the rows (4 x 16 TN rows): the wide layers (10^4 ... 10^5 rows), bound by the row traffic;
//   KSPLIT = 2/4 the waves (also) split the contraction -- wave kp takes the k-steps kp, kp + KSPLIT, ... -- and their accumulators are added
//                through LDS in wave order: the coarse encoder levels (a few hundred rows against 2048 ... 8192 input channels), whose grid is a
//                few dozen workgroups and whose time is the LENGTH of the dependent load chain, not bytes.
// PF k-steps of fragments are in flight per wave (a ring of PF register sets).
// With TM = 4 the A rows are taken in the order  MFMA row 4 g + r of block i  <->  channel 16 g + 4 i + r  of the 64-channel block, so that lane
// (n, g) ends up with the 16 CONSECUTIVE channels 16 g .. 16 g + 15 of row n: a row's 64 channels leave as 128 contiguous bytes (with the plain
// order a lane stores 8 bytes per block and a store instruction touches 16 rows x 32 bytes).
// ---------------------------------------------------------------------------------------------------------------------
struct NtArgs {
    const uint16_t* x; int64_t ldx;
    const uint16_t* w; int64_t ldw;
    const float* bias;
    void* y; int64_t ldy;
    int64_t m; int n, k;
    int out_f32;
};

template <int TM, int TN, int KSPLIT, int PF, bool F16>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const NtArgs a) {
    constexpr int RG = 4 / KSPLIT;                                          // row groups of a workgroup
    constexpr bool PERM = TM == 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i16 = lane & 15, kg = lane >> 4;
    const int rg = wave / KSPLIT, kp = wave % KSPLIT;
    const int64_t r0 = ((int64_t)blockIdx.x * RG + rg) * (16 * TN);
    const int c0 = blockIdx.y * (16 * TM);
    const bool rows_live = r0 < a.m;                                        // (no early return: the waves of a split contraction meet at a barrier)
    const uint16_t* xr[TN];
    const uint16_t* wr[TM];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int64_t row = r0 + 16 * t + i16;
        xr[t] = a.x + (row < a.m ? row : a.m - 1) * a.ldx + 8 * kg;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ch = c0 + (PERM ? 16 * (i16 >> 2) + 4 * i + (i16 & 3) : 16 * i + i16);
        wr[i] = a.w + (int64_t)(ch < a.n ? ch : a.n - 1) * a.ldw + 8 * kg;
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ksteps = (a.k + 31) / 32;
    const int mine = (ksteps - kp + KSPLIT - 1) / KSPLIT;                    // k-steps kp, kp + KSPLIT, ... of this wave
    u32x4 fa[PF][TM], fb[PF][TN];
    const u32x4 zero = {0, 0, 0, 0};
    auto load = [&](int i, u32x4 (&pa)[TM], u32x4 (&pb)[TN]) {
        const int s = kp + KSPLIT * i;
        const bool in = rows_live && i < mine && 32 * s + 8 * kg < a.k;      // K is a multiple of 8: a chunk is inside or outside as a whole
#pragma unroll
        for (int j = 0; j < TM; ++j) pa[j] = in ? *(const u32x4*)(wr[j] + 32 * s) : zero;
#pragma unroll
        for (int t = 0; t < TN; ++t) pb[t] = in ? *(const u32x4*)(xr[t] + 32 * s) : zero;
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) load(p, fa[p], fb[p]);
    for (int i0 = 0; i0 < mine; i0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int t = 0; t < TN; ++t) acc[j][t] = mfma16<F16>(fa[p][j], fb[p][t], acc[j][t]);
            load(i0 + p + PF, fa[p], fb[p]);
        }
    }
    if constexpr (KSPLIT > 1) {
        __shared__ float red[RG][KSPLIT - 1][TM * TN * 4][64];
        if (kp > 0) {
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int t = 0; t < TN; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[rg][kp - 1][(j * TN + t) * 4 + r][lane] = acc[j][t][r];
        }
        __syncthreads();
        if (kp > 0) return;
#pragma unroll
        for (int q = 0; q < KSPLIT - 1; ++q)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int t = 0; t < TN; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[j][t][r] += red[rg][q][(j * TN + t) * 4 + r][lane];
    }
    if (!rows_live) return;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int64_t row = r0 + 16 * t + i16;
        if (row >= a.m) continue;
        if constexpr (PERM) {
            const int ch = c0 + 16 * kg;                                     // this lane: channels ch .. ch + 15, value (i, r) = channel ch + 4 i + r
            if (ch >= a.n) continue;
            float v[16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * i + r] = acc[i][t][r] + ((a.bias && ch + 4 * i + r < a.n) ? a.bias[ch + 4 * i + r] : 0.f);
            if (a.out_f32) {
                float* dst = (float*)a.y + row * a.ldy + ch;
                if (ch + 15 < a.n && (a.ldy & 3) == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) *(f32x4*)(dst + 4 * i) = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
                } else {
                    for (int e = 0; e < 16 && ch + e < a.n; ++e) dst[e] = v[e];
                }
            } else {
                uint16_t* dst = (uint16_t*)a.y + row * a.ldy + ch;
                unsigned p[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) p[e] = pack2<F16>(v[2 * e], v[2 * e + 1]);
                if (ch + 15 < a.n && (a.ldy & 7) == 0) {
                    *(u32x4*)dst = u32x4{p[0], p[1], p[2], p[3]};
                    *(u32x4*)(dst + 8) = u32x4{p[4], p[5], p[6], p[7]};
                } else {
                    for (int e = 0; e < 16 && ch + e < a.n; ++e) dst[e] = (uint16_t)(e & 1 ? p[e >> 1] >> 16 : p[e >> 1] & 0xffff);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ch = c0 + 16 * i + 4 * kg;
                if (ch >= a.n) continue;
                f32x4 v = acc[i][t];
                if (a.bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (ch + r < a.n) ? a.bias[ch + r] : 0.f;
                }
                if (a.out_f32) {
                    float* dst = (float*)a.y + row * a.ldy + ch;
                    if (ch + 3 < a.n && (a.ldy & 3) == 0) *(f32x4*)dst = v;
                    else for (int r = 0; r < 4 && ch + r < a.n; ++r) dst[r] = v[r];
                } else {
                    uint16_t* dst = (uint16_t*)a.y + row * a.ldy + ch;
                    const unsigned p0 = pack2<F16>(v[0], v[1]), p1 = pack2<F16>(v[2], v[3]);
                    if (ch + 3 < a.n && (a.ldy & 3) == 0) *(u32x2*)dst = u32x2{p0, p1};
                    else {
                        const uint16_t h[4] = {(uint16_t)(p0 & 0xffff), (uint16_t)(p0 >> 16), (uint16_t)(p1 & 0xffff), (uint16_t)(p1 >> 16)};
                        for (int r = 0; r < 4 && ch + r < a.n; ++r) dst[r] = h[r];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// TN product: dw[co][ci] = sum_rows g[row][co] x[row][ci].  Workgroup = 64 co x 64 ci over a slab of rows; wave (wm, wn) = 32 x 32.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TN_T = 64;                 // tile edge (channels) and rows per staged step
constexpr int TN_PITCH = TN_T + 16;      // LDS row pitch in elements: + 32 bytes keeps the transpose reads of a half-wave on distinct banks

struct TnArgs {
    const uint16_t* g; int64_t ldg;
    const uint16_t* x; int64_t ldx;
    float* part;                          // [slabs][n][k]
    int64_t m; int n, k;
    int64_t rows_per_slab;                // multiple of TN_T
    int ci_tiles;
};

__device__ __forceinline__ u32x2 lds_tr_read16(const uint16_t* p) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)p) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait(u32x2& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory"); }

template <bool F16>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const TnArgs a) {
    __shared__ __attribute__((aligned(16))) uint16_t gimg[2][TN_T * TN_PITCH], ximg[2][TN_T * TN_PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i16 = lane & 15, kg = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const int co0 = (blockIdx.x / a.ci_tiles) * TN_T, ci0 = (blockIdx.x % a.ci_tiles) * TN_T;
    const int64_t row_begin = (int64_t)blockIdx.y * a.rows_per_slab;
    const int64_t row_end = row_begin + a.rows_per_slab < a.m ? row_begin + a.rows_per_slab : a.m;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging: 64 rows x 64 channels = 512 chunks of 16 bytes per operand, two per thread
    const int srow[2] = {(int)(threadIdx.x >> 3), (int)(threadIdx.x >> 3) + 32};
    const int sch = threadIdx.x & 7;
    const bool g_in = co0 + 8 * sch < a.n, x_in = ci0 + 8 * sch < a.k;         // N and K are multiples of 8: a chunk is inside or outside as a whole
    u32x4 rg[2], rx[2];
    const u32x4 zero = {0, 0, 0, 0};
    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t row = r0 + srow[h];
            const bool live = row < row_end;
            rg[h] = (live && g_in) ? *(const u32x4*)(a.g + row * a.ldg + co0 + 8 * sch) : zero;
            rx[h] = (live && x_in) ? *(const u32x4*)(a.x + row * a.ldx + ci0 + 8 * sch) : zero;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *(u32x4*)(gimg[buf] + srow[h] * TN_PITCH + 8 * sch) = rg[h];
            *(u32x4*)(ximg[buf] + srow[h] * TN_PITCH + 8 * sch) = rx[h];
        }
    };
    const int prow = 4 * kg + (i16 >> 2), pcol = 4 * (i16 & 3);
    auto products = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < TN_T / 32; ++ks) {
            const uint16_t* gs = gimg[buf] + 32 * ks * TN_PITCH;
            const uint16_t* xs = ximg[buf] + 32 * ks * TN_PITCH;
            u32x2 af[2][2], bf[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i][0] = lds_tr_read16(gs + prow * TN_PITCH + 16 * (2 * wm + i) + pcol);
                af[i][1] = lds_tr_read16(gs + (16 + prow) * TN_PITCH + 16 * (2 * wm + i) + pcol);
                bf[i][0] = lds_tr_read16(xs + prow * TN_PITCH + 16 * (2 * wn + i) + pcol);
                bf[i][1] = lds_tr_read16(xs + (16 + prow) * TN_PITCH + 16 * (2 * wn + i) + pcol);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) { lds_wait(af[i][0]); lds_wait(af[i][1]); lds_wait(bf[i][0]); lds_wait(bf[i][1]); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = mfma16<F16>(u32x4{af[i][0].x, af[i][0].y, af[i][1].x, af[i][1].y}, u32x4{bf[j][0].x, bf[j][0].y, bf[j][1].x, bf[j][1].y}, acc[i][j]);
        }
    };
    int buf = 0;
    if (row_begin < row_end) fetch(row_begin);
    for (int64_t r0 = row_begin; r0 < row_end; r0 += TN_T) {
        stage(buf);
        if (r0 + TN_T < row_end) fetch(r0 + TN_T);
        __syncthreads();                                            // (also: the products of the previous step, on the other buffer, are done)
        products(buf);
        buf ^= 1;
    }
    float* out = a.part + (int64_t)blockIdx.y * a.n * a.k;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + 16 * (2 * wm + i) + 4 * kg + r, ci = ci0 + 16 * (2 * wn + j) + i16;
                if (co < a.n && ci < a.k) out[(int64_t)co * a.k + ci] = acc[i][j][r];
            }
}

// out[i] = sum over the slabs: 16 threads share an element (slab = slice, slice + 16, ... in double), their sums are added in slice order -- the
// result does not depend on scheduling.  (One thread per element walked all slabs alone: 390 dependent loads for the 100 000-row layers, 24 us.)
__global__ __launch_bounds__(256) void tn_sum_kernel(const float* __restrict__ part, int slabs, int64_t n, float* __restrict__ out) {
    __shared__ double red[16][16];
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + e;
    double s = 0.0;
    if (i < n) {
#pragma unroll 4
        for (int p = sl; p < slabs; p += 16) s += (double)part[(int64_t)p * n + i];
    }
    red[sl][e] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][e];
        out[i] = (float)t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// transposed 16-bit images of fp32 matrices: table of (src [n, k] fp32, dst [k, n] 16-bit, n, k, first 32 x 32 tile) entries, one workgroup per tile
// ---------------------------------------------------------------------------------------------------------------------
struct TrEntry { const float* src; uint16_t* dst; int n, k; int64_t tile0; };

template <bool F16>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const TrEntry* __restrict__ table, int entries) {
    __shared__ float tile[32][33];
    int lo = 0, hi = entries - 1;                                   // the entry whose tile range holds this block (tile0 ascending)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].tile0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const TrEntry e = table[lo];
    const int tk = (e.k + 31) / 32;
    const int64_t t = (int64_t)blockIdx.x - e.tile0;
    const int n0 = (int)(t / tk) * 32, k0 = (int)(t % tk) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = (n0 + r < e.n && k0 + tx < e.k) ? e.src[(int64_t)(n0 + r) * e.k + k0 + tx] : 0.f;
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        if (k0 + r < e.k && n0 + tx < e.n) {
            const unsigned p = pack2<F16>(tile[tx][r], 0.f);
            e.dst[(int64_t)(k0 + r) * e.n + n0 + tx] = (uint16_t)(p & 0xffff);
        }
    }
}

template <bool F16>
int launch_nt(const NtArgs& a, hipStream_t st) {
    if ((a.m + 31) / 32 > 0x7fffffff) return PPS_ERR_ARG;
    const int ksteps = (a.k + 31) / 32;
    if (a.n > 32) {
        // 64-channel blocks.  The contraction is split between the waves while the grid would not fill the chip and every wave keeps >= 4 k-steps
        const int64_t cb = (a.n + 63) / 64;
        int ks = 1;
        while (ks < 4 && ((a.m + 128 / ks - 1) / (128 / ks)) * cb < 512 && ksteps / (2 * ks) >= 4) ks *= 2;
        const dim3 grid((unsigned)((a.m + 128 / ks - 1) / (128 / ks)), (unsigned)cb);
        if (ks == 1) hipLaunchKernelGGL((gemm_nt_kernel<4, 2, 1, 2, F16>), grid, dim3(256), 0, st, a);
        else if (ks == 2) hipLaunchKernelGGL((gemm_nt_kernel<4, 2, 2, 4, F16>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_nt_kernel<4, 2, 4, 4, F16>), grid, dim3(256), 0, st, a);
    } else if (a.n > 16) {
        hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 1, 2, F16>), dim3((unsigned)((a.m + 127) / 128), 1), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((gemm_nt_kernel<1, 2, 1, 2, F16>), dim3((unsigned)((a.m + 127) / 128), 1), dim3(256), 0, st, a);
    }
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int tn_slabs(int64_t m, int n, int k) {
    // enough workgroups to fill the chip, few enough that the partials stay small: tiles x slabs ~ 512, at least 512 rows per slab, at most 64 MB
    const int64_t tiles = (int64_t)((n + TN_T - 1) / TN_T) * ((k + TN_T - 1) / TN_T);
    int64_t slabs = (512 + tiles - 1) / tiles;
    const int64_t by_rows = (m + 511) / 512;
    if (slabs > by_rows) slabs = by_rows;
    const int64_t by_bytes = ((int64_t)64 << 20) / ((int64_t)n * k * 4);
    if (slabs > by_bytes) slabs = by_bytes;
    if (slabs < 1) slabs = 1;
    if (slabs > 1024) slabs = 1024;
    return (int)slabs;
}

}  // namespace

extern "C" {

/* y [m, n] = x [m, k] w [n, k]^T (+ bias [n] fp32).  x, w: 16-bit (dtype 1 bfloat16, 2 IEEE half) with row pitches ldx, ldw (elements); y 16-bit of
 * the same type, or fp32 if out_f32; k and the pitches multiples of 8, pointers 16-byte aligned.  The input gradient of the layer is the same call
 * with (g, transposed image of w). */
int pps_gemm_nt_16(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, void* y, int64_t ldy, int64_t m, int n, int k, int dtype,
                   int out_f32, void* stream) {
    if (m < 0 || n < 1 || k < 8 || (k & 7) || (ldx & 7) || (ldw & 7) || ldx < k || ldw < k || ldy < n || (dtype != 1 && dtype != 2)) return PPS_ERR_ARG;
    if (m == 0) return PPS_OK;
    if (!x || !w || !y) return PPS_ERR_ARG;
    NtArgs a{(const uint16_t*)x, ldx, (const uint16_t*)w, ldw, bias, y, ldy, m, n, k, out_f32};
    return dtype == 2 ? launch_nt<true>(a, (hipStream_t)stream) : launch_nt<false>(a, (hipStream_t)stream);
}

/* scratch of pps_gemm_tn_16 in bytes */
size_t pps_gemm_tn_ws_bytes(int64_t m, int n, int k) { return (m < 1 || n < 1 || k < 1) ? 0 : (size_t)tn_slabs(m, n, k) * n * k * sizeof(float); }

/* dw [n, k] fp32 = g [m, n]^T x [m, k]: g, x 16-bit with row pitches ldg, ldx (multiples of 8 elements; n and k multiples of 8); ws:
 * pps_gemm_tn_ws_bytes(m, n, k) bytes.  Slabs of rows are summed in slab order: the result does not depend on scheduling. */
int pps_gemm_tn_16(const void* g, int64_t ldg, const void* x, int64_t ldx, int64_t m, int n, int k, int dtype, float* dw, void* ws, void* stream) {
    if (m < 1 || n < 8 || k < 8 || (n & 7) || (k & 7) || (ldg & 7) || (ldx & 7) || ldg < n || ldx < k || (dtype != 1 && dtype != 2)) return PPS_ERR_ARG;
    if (!g || !x || !dw || !ws) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int slabs = tn_slabs(m, n, k);
    int64_t per = ((m + slabs - 1) / slabs + TN_T - 1) / TN_T * TN_T;
    const int used = (int)((m + per - 1) / per);
    TnArgs a{(const uint16_t*)g, ldg, (const uint16_t*)x, ldx, used == 1 ? dw : (float*)ws, m, n, k, per, (k + TN_T - 1) / TN_T};
    const dim3 grid((unsigned)(((n + TN_T - 1) / TN_T) * a.ci_tiles), (unsigned)used);
    if (dtype == 2) hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(256), 0, st, a);
    if (used > 1) {
        const int64_t nk = (int64_t)n * k;
        hipLaunchKernelGGL(tn_sum_kernel, dim3((unsigned)((nk + 15) / 16)), dim3(256), 0, st, (const float*)ws, used, nk, dw);
    }
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

/* Transposed 16-bit images of many fp32 matrices in one launch.  table: device array of `entries` records {const float* src [n, k]; void* dst [k, n];
 * int32 n, k; int64 tile0} (pps_transpose_entry_bytes() bytes each), tile0 = number of 32 x 32 tiles of the entries before this one; tiles = total. */
int pps_transpose_entry_bytes(void) { return (int)sizeof(TrEntry); }

int pps_transpose_cast_pieces(const void* table, int entries, int64_t tiles, int dtype, void* stream) {
    if (entries < 0 || tiles < 0 || tiles > 0x7fffffff || (dtype != 1 && dtype != 2)) return PPS_ERR_ARG;
    if (entries == 0 || tiles == 0) return PPS_OK;
    if (!table) return PPS_ERR_ARG;
    if (dtype == 2) hipLaunchKernelGGL(transpose_cast_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, (const TrEntry*)table, entries);
    else hipLaunchKernelGGL(transpose_cast_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, (const TrEntry*)table, entries);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

}  // extern "C"
