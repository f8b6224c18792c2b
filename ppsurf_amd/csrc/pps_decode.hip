// Occupancy decoder of PPSurf for gfx950: interpolation-attention branch, PointNet patch branch, MLP tail.
//
// All kernels share one structure (pps_common.h):
//   * a workgroup = 8 waves (512 threads, 2 waves per SIMD, <= 256 VGPRs), persistent over "tiles";
//   * each wave owns 16 rows and keeps their activations in registers in the MFMA C/D layout of
//     v_mfma_f32_16x16x4_f32 (exact fp32, 157 TFLOP/s peak), chaining layers without leaving registers;
//   * the layer weights, packed on the host in A-operand order, are streamed global(L2) -> registers -> LDS
//     in 8..32 KiB chunks, double buffered, one barrier per chunk, shared by the 8 waves.
//
// Reference semantics (eval mode, BatchNorm folded by the host, see ppsurf_amd/decoder.py):
//   source/poco_model.py:381-419, source/base/nn.py:72-96,133-190,305-373,415-417, source/ppsurf_model.py:82-117.
#include <cstdlib>
#include "pps_common.h"
#include "../../include/ppsurf_amd.h"

using namespace pps;

#ifndef NT
#define NT 256                 // threads per workgroup (4 waves); 2 workgroups per CU run decoupled
#endif
#define NW (NT / 64)            // waves per workgroup
#ifndef WG_PER_CU
#define WG_PER_CU (512 / NT)
#endif
#ifndef PPS_PRIO
#define PPS_PRIO 1
#endif
#ifndef PPS_PRIO_PN
#define PPS_PRIO_PN 1
#endif
#define CH4 2048               // f32x4 per 32 KiB weight chunk

// One pipeline step: request the NEXT chunk, compute on the CURRENT one, publish the next, barrier.
// CHUNK_F4_NEXT: size of the next chunk in f32x4; NTH: threads of the workgroup (all of them copy)
template <int CHUNK_F4_NEXT, int NTH = NT, class F>
__device__ __forceinline__ void stream_step(const f32x4* __restrict__ gnext, f32x4*& cur, f32x4*& nxt, F&& compute) {
    chunk_copy_async<CHUNK_F4_NEXT / NTH, NTH>(gnext, nxt);
    compute((const f32x4*)cur);
    stream_wait();
    __syncthreads();
    f32x4* t = cur; cur = nxt; nxt = t;
}

// The same pipeline step with the copy of the next chunk SPREAD over the compute phase: `compute(w, piece)` calls piece(i), i = 0 .. NPIECES-1, at
// points of its own choosing (dense_blocks_f16x3_hook: after the MFMAs of a k-step).  Issued in one burst at the head of the step, the pieces of all
// waves of a workgroup queue up in the CU's memory pipeline (32 KiB at 64 B/clk = 512 cycles per chunk) while every wave sits in the issue and the
// matrix pipe is idle -- a cycle trace of the split-precision interpolation kernel (round 3, profiles/NOTES_r3.md) showed 12 % of a step in the issue
// and 20 % in the barrier behind it; spread out, a piece is accepted while the wave's own queued MFMAs execute.
template <int CHUNK_F4_NEXT, int NTH = NT, class F>
__device__ __forceinline__ void stream_step_spread(const f32x4* __restrict__ gnext, f32x4*& cur, f32x4*& nxt, F&& compute) {
    f32x4* dst = nxt;
    compute((const f32x4*)cur, [&](int i) { if (i < CHUNK_F4_NEXT / NTH) chunk_copy_piece<NTH>(gnext, dst, i); });
    stream_wait();
    __syncthreads();
    f32x4* t = cur; cur = nxt; nxt = t;
}

// stream_step_spread with the pieces addressed by chunk_copy_piece_at: `chunk` = first byte of the NEXT chunk (SGPR pair kept by the caller),
// voff = the per-lane piece offsets of stream_lane_offsets (no vector instruction per piece)
template <int CHUNK_F4_NEXT, int NTH = NT, class F>
__device__ __forceinline__ void stream_step_spread_at(const char* chunk, const unsigned (&voff)[CHUNK_F4_NEXT / NTH], f32x4*& cur, f32x4*& nxt, F&& compute) {
    f32x4* dst = nxt;
    compute((const f32x4*)cur, [&](int i) { if (i < CHUNK_F4_NEXT / NTH) chunk_copy_piece_at<NTH>(chunk, dst, i, voff[i]); });
    stream_wait();
    __syncthreads();
    f32x4* t = cur; cur = nxt; nxt = t;
}

template <int CHUNK_F4, int NTH = NT>
__device__ __forceinline__ void stream_prologue(const f32x4* __restrict__ g, f32x4* buf) {
    chunk_copy_async<CHUNK_F4 / NTH, NTH>(g, buf);
}

__device__ __forceinline__ void lds_fill(float* dst, const float* __restrict__ src, int nfloats) {
    for (int i = threadIdx.x; i < nfloats; i += blockDim.x) dst[i] = src[i];
}

// reductions over aligned groups of LO in {1,2,4,8} adjacent rows (lanes n): every lane of a group gets the result
__device__ __forceinline__ float group_max(float v, int lo) {
    if (lo >= 2) v = max_raw(v, dpp_mov<0xB1>(v));         // (max_raw: one v_max_f32, no quieting moves in front; pps_common.h)
    if (lo >= 4) v = max_raw(v, dpp_mov<0x4E>(v));
    if (lo >= 8) v = max_raw(v, dpp_mov<0x141>(v));
    return v;
}
__device__ __forceinline__ float group_sum(float v, int lo) {
    if (lo >= 2) v += dpp_mov<0xB1>(v);
    if (lo >= 4) v += dpp_mov<0x4E>(v);
    if (lo >= 8) v += dpp_mov<0x141>(v);
    return v;
}

template <int N>
__device__ __forceinline__ void relu_blocks(f32x4* a) {
#pragma unroll
    for (int b = 0; b < N; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[b][r] = fmaxf(a[b][r], 0.f);
}

extern __shared__ __attribute__((aligned(16))) char pps_smem[];


// =====================================================================================================
// rows_dense256: out[m,256] = in[m,256] W^T + b
// =====================================================================================================
#define RD_LDS_BYTES (2 * CH4 * 16 + 256 * 4)

__global__ __launch_bounds__(NT, 2) void rows_dense256_kernel(const float* __restrict__ in, int64_t rs, int64_t cs, int64_t m,
                                                              const float* __restrict__ wpack, const float* __restrict__ bias,
                                                              float* __restrict__ out) {
    f32x4* buf0 = (f32x4*)pps_smem;
    f32x4* buf1 = buf0 + CH4;
    float* bias_l = (float*)(buf1 + CH4);
    const f32x4* bias4 = (const f32x4*)bias_l;
    const f32x4* wg = (const f32x4*)wpack;
    const int lane = lane_id(), wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;

    lds_fill(bias_l, bias, 256);
    stream_prologue<CH4>(wg, buf0);
    stream_wait();
    __syncthreads();
    f32x4 *cur = buf0, *nxt = buf1;

    const int ntiles = (int)((m + NW * 16 - 1) / (NW * 16));
    int first, count, stride;
    xcd_tile_range(ntiles, first, count, stride);
    for (int it = 0; it < count; ++it) {
        const int64_t row = (int64_t)(first + it * stride) * (NW * 16) + wave * 16 + n;
        const bool rv = row < m;
        const int64_t rc = rv ? row : m - 1;
        f32x4 a[16];
        if (cs == 1) {
            const f32x4* src = (const f32x4*)(in + rc * rs) + g;
#pragma unroll
            for (int b = 0; b < 16; ++b) a[b] = src[4 * b];
        } else {
#pragma unroll
            for (int b = 0; b < 16; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) a[b][r] = in[rc * rs + (int64_t)(16 * b + 4 * g + r) * cs];
        }
        f32x4* dst = (f32x4*)(out + rc * 256) + g;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            f32x4 o[2];
            stream_step<CH4>(wg + ((c + 1) & 7) * CH4, cur, nxt,
                           [&](const f32x4* w) { dense_blocks<16, 2, 0>(a, o, w, bias4 + 8 * c, lane); });
            if (rv) { dst[4 * (2 * c)] = o[0]; dst[4 * (2 * c + 1)] = o[1]; }
        }
    }
}

// =====================================================================================================
// interp_pool: gather(G, xyz) -> +fc1_xyz, ReLU -> fc2 -> fc3 -> fc_query -> softmax_j, mean_heads -> sum_j a_j h3_j
// weights (floats): [xyz 1024][fc2 65536][fc3 65536][fcq 16384]   bias: [256][256][64]
// =====================================================================================================
#define IP_W_XYZ 1024
#define IP_NBIAS 576
// LDS floats: xyz 1024 | bias 576 | ms_m NW*64 | ms_s NW*64 | f NW*64 | part NW*256
#define IP_LDS_BYTES (2 * CH4 * 16 + (IP_W_XYZ + IP_NBIAS + NW * 64 * 3 + NW * 256) * 4)

__global__ __launch_bounds__(NT, 2) void interp_pool_kernel(const float* __restrict__ G, const float* __restrict__ pts,
                                                            const float* __restrict__ query, const int64_t* __restrict__ idx,
                                                            int64_t Q, int k, const float* __restrict__ wpack,
                                                            const float* __restrict__ bias, float* __restrict__ pooled,
                                                            const int* __restrict__ gate) {
    if (gate_closed(gate)) return;
    f32x4* buf0 = (f32x4*)pps_smem;
    f32x4* buf1 = buf0 + CH4;
    float* xyz_l = (float*)(buf1 + CH4);
    float* bias_l = xyz_l + IP_W_XYZ;
    float* msm = bias_l + IP_NBIAS;      // [8][64] per-wave row max of every head
    float* mss = msm + NW * 64;          // [NW][64] per-wave sum of exp
    float* f_l = mss + NW * 64;          // [NW][64] per-wave head factor exp(m_w - M) / (64 S)
    float* part = f_l + NW * 64;         // [NW][256] per-wave pooled partial
    const f32x4* bias4 = (const f32x4*)bias_l;
    const f32x4* wg = (const f32x4*)(wpack + IP_W_XYZ);
    const int lane = lane_id(), wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;

    lds_fill(xyz_l, wpack, IP_W_XYZ);
    lds_fill(bias_l, bias, IP_NBIAS);
    stream_prologue<CH4>(wg, buf0);
    stream_wait();
    __syncthreads();
    f32x4 *cur = buf0, *nxt = buf1;

    const int ntiles = (int)((Q + NW / 4 - 1) / (NW / 4));       // 4 waves (64 neighbours) per query
    int first, count, stride;
    xcd_tile_range(ntiles, first, count, stride);
    const int wq = wave & 3, wbase = wave & ~3;
    for (int it = 0; it < count; ++it) {
        const int64_t qi = (int64_t)(first + it * stride) * (NW / 4) + (wave >> 2);
        const bool qv = qi < Q;
        const int64_t qc = qv ? qi : Q - 1;
        const int row = wq * 16 + n;
        const bool valid = row < k;
        const int64_t i = idx[qc * k + (valid ? row : 0)];

        f32x4 a[16], b[16];
        {
            const f32x4* grow = (const f32x4*)(G + i * 256) + g;
#pragma unroll
            for (int bb = 0; bb < 16; ++bb) a[bb] = grow[4 * bb];
            const float coord = (g < 3) ? (query[qc * 3 + g] - pts[i * 3 + g]) : 0.f;   // query minus neighbour (poco_model.py:402)
            xyz_blocks<16>(coord, a, xyz_l, lane);
            relu_blocks<16>(a);
        }
        // two waves share a SIMD: the one in its MFMA phase gets the issue slots first, so the matrix pipe is not left idle
        // behind the other wave's gather / softmax VALU work (+1.8 % measured)
        __builtin_amdgcn_s_setprio(PPS_PRIO);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            stream_step<CH4>(wg + (c + 1) * CH4, cur, nxt,
                           [&](const f32x4* w) { dense_blocks<16, 2, 1>(a, &b[2 * c], w, bias4 + 8 * c, lane); });
#pragma unroll
        for (int c = 0; c < 8; ++c)
            stream_step<CH4>(wg + (c + 9) * CH4, cur, nxt,
                           [&](const f32x4* w) { dense_blocks<16, 2, 1>(b, &a[2 * c], w, bias4 + 64 + 8 * c, lane); });
#pragma unroll
        for (int c = 0; c < 2; ++c)
            stream_step<CH4>(wg + ((c + 17) % 18) * CH4, cur, nxt,
                           [&](const f32x4* w) { dense_blocks<16, 2, 0>(a, &b[2 * c], w, bias4 + 128 + 8 * c, lane); });

        __builtin_amdgcn_s_setprio(0);
        // ---- softmax over the 64 neighbours (4 waves x 16 rows) for each of the 64 heads -------------
        // lane (n,g) holds heads 16*bb + 4*g + r of row n in b[bb][r]
        float e[16];
        {
            f32x4 m4[4], s4[4];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = valid ? b[bb][r] : -INFINITY;
                    const float mx = row16_max(v);
                    const float ev = valid ? __expf(v - mx) : 0.f;
                    e[bb * 4 + r] = ev;
                    m4[bb][r] = mx;
                    s4[bb][r] = row16_sum(ev);
                }
            if (n < 4) {
                const f32x4 mm = (n == 0) ? m4[0] : (n == 1) ? m4[1] : (n == 2) ? m4[2] : m4[3];
                const f32x4 ss = (n == 0) ? s4[0] : (n == 1) ? s4[1] : (n == 2) ? s4[2] : s4[3];
                ((f32x4*)(msm + wave * 64))[4 * n + g] = mm;
                ((f32x4*)(mss + wave * 64))[4 * n + g] = ss;
            }
        }
        __syncthreads();
        {
            // lane = head: combine the 4 waves of this query
            float mw[4], sw[4];
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) { mw[w2] = msm[(wbase + w2) * 64 + lane]; sw[w2] = mss[(wbase + w2) * 64 + lane]; }
            const float M = fmaxf(fmaxf(mw[0], mw[1]), fmaxf(mw[2], mw[3]));
            float S = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) S += sw[w2] * __expf(mw[w2] - M);
            f_l[wave * 64 + lane] = __expf(mw[wq] - M) / (64.f * S);
        }
        __syncthreads();
        float an = 0.f;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x4 f4 = ((const f32x4*)(f_l + wave * 64))[4 * bb + g];
#pragma unroll
            for (int r = 0; r < 4; ++r) an += e[bb * 4 + r] * f4[r];
        }
        an += __shfl_xor(an, 16);
        an += __shfl_xor(an, 32);           // attention weight of row n (mean over the 64 heads)
#pragma unroll
        for (int bb = 0; bb < 16; ++bb) a[bb] *= an;
        rows16_sum_transposed(a, lane);                     // lane (n,g): a[0] = sum over the 16 rows of block n
        ((f32x4*)(part + wave * 256))[4 * n + g] = a[0];
        __syncthreads();
        if (wq == 0 && qv) {
            f32x4 s = ((const f32x4*)(part + (wbase + 0) * 256))[lane];
            s += ((const f32x4*)(part + (wbase + 1) * 256))[lane];
            s += ((const f32x4*)(part + (wbase + 2) * 256))[lane];
            s += ((const f32x4*)(part + (wbase + 3) * 256))[lane];
            ((f32x4*)(pooled + qi * 256))[lane] = s;
        }
    }
}

// =====================================================================================================
// interp_pool_f16x3: the same branch with the three big layers (fc2, fc3, fc_query) on the f16 matrix pipe in split precision
// (pps_common.h, dense_blocks_f16x3) -- opt-in decoder dtype "f16x3"; gather, the xyz part of fc1, softmax and pooling stay fp32.
// One workgroup = 8 waves = 2 queries (128 rows) per pass over the 576 KB of fc2 / fc3 / fc_query.  What bounds it: not the
// weight stream and not the LDS port as such -- a bare loop of the same MFMAs on random operands sustains 1.8 PFLOP/s (the chip clocks down under
// f16 matrix load, tools/ubench/mfma_power_probe.hip), 1.7 with the A fragments re-read from LDS; this kernel reaches 1.1.
// weights: wxyz (floats) [xyz 1024]; w16 (half8 fragments) [fc2 16 ob x 8 kb][fc3 16 x 8][fcq 4 x 8], 2 KiB per (ob, kb);
// bias (floats): [256][256][64]
// =====================================================================================================
#define IH_NT 512              // one 8-wave workgroup per CU: 2 queries (128 rows) per pass over the weights
#define IH_OB 2                // output blocks per streamed weight chunk: 32 KiB chunks, 18 per pass
#define IH_NW (IH_NT / 64)
#define IH_WG_PER_CU (512 / IH_NT)
#define IH_CH4 (CH4 * IH_OB / 2)
#define IH_NCH (32 / IH_OB + 4 / IH_OB)       // chunks per pass: fc2 and fc3 16 output blocks each, fc_query 4
#define IH_LDS_BYTES (2 * IH_CH4 * 16 + (IP_W_XYZ + IP_NBIAS + IH_NW * 64 * 3 + IH_NW * 256) * 4)

__global__ __launch_bounds__(IH_NT, 2) void interp_pool_f16x3_kernel(const float* __restrict__ G, const float* __restrict__ pts,
                                                                     const float* __restrict__ query, const int64_t* __restrict__ idx,
                                                                     int64_t Q, int k, const float* __restrict__ wxyz,
                                                                     const f32x4* __restrict__ w16, const float* __restrict__ bias,
                                                                     float* __restrict__ pooled, int* __restrict__ range) {
    float amax = 0.f;
    f32x4* buf0 = (f32x4*)pps_smem;
    f32x4* buf1 = buf0 + IH_CH4;
    float* xyz_l = (float*)(buf1 + IH_CH4);
    float* bias_l = xyz_l + IP_W_XYZ;
    float* msm = bias_l + IP_NBIAS;
    float* mss = msm + IH_NW * 64;
    float* f_l = mss + IH_NW * 64;
    float* part = f_l + IH_NW * 64;
    const f32x4* bias4 = (const f32x4*)bias_l;
    const f32x4* wg = w16;
    const int lane = lane_id(), wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;

    lds_fill(xyz_l, wxyz, IP_W_XYZ);
    lds_fill(bias_l, bias, IP_NBIAS);
    stream_prologue<IH_CH4, IH_NT>(wg, buf0);
    stream_wait();
    __syncthreads();
    f32x4 *cur = buf0, *nxt = buf1;
    // the weight stream: one SGPR pointer to the chunk being fetched (advanced after every step, back to the first chunk after the last one) and
    // one VGPR byte offset per piece -- no vector instruction and no spilled pointer per piece (pps_common.h, chunk_copy_piece_at)
    unsigned voff[IH_CH4 / IH_NT];
    stream_lane_offsets<IH_NT>(voff);
    const char* const wfirst = (const char*)wg;
    const char* snext = wfirst + (size_t)IH_CH4 * 16;

    const int ntiles = (int)((Q + IH_NW / 4 - 1) / (IH_NW / 4));
    int first, count, stride;
    xcd_tile_range(ntiles, first, count, stride);
    const int wq = wave & 3, wbase = wave & ~3;
    for (int it = 0; it < count; ++it) {
        const int64_t qi = (int64_t)(first + it * stride) * (IH_NW / 4) + (wave >> 2);
        const bool qv = qi < Q;
        const int64_t qc = qv ? qi : Q - 1;
        const int row = wq * 16 + n;
        const bool valid = row < k;
        const int64_t i = idx[qc * k + (valid ? row : 0)];

        HiLo x[8], y[8];
        {
            f32x4 a[16];
            const f32x4* grow = (const f32x4*)(G + i * 256) + g;
#pragma unroll
            for (int bb = 0; bb < 16; ++bb) a[bb] = grow[4 * bb];
            const float coord = (g < 3) ? (query[qc * 3 + g] - pts[i * 3 + g]) : 0.f;   // query minus neighbour (poco_model.py:402)
            xyz_blocks<16>(coord, a, xyz_l, lane);
            relu_blocks<16>(a);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) x[kb] = split_f16_r(amax, a[2 * kb], a[2 * kb + 1]);
        }
        __builtin_amdgcn_s_setprio(PPS_PRIO);
        // the 4 (8-wave workgroup) pieces of the next chunk's copy go out after k-steps 0, 2, 4, 6 of the first output-block pair
#ifndef PPS_FCQ_PRODUCTS
#define PPS_FCQ_PRODUCTS 3      // f16 products per fp32 product in fc_query (2, 1: experiment builds only, python -m ppsurf_amd.build --variant)
#endif
        // LAST: the step that fetches the last chunk of the pass -- the pointer wraps to the first chunk behind it
#define IH_STEP(LAST, IN, BIAS, ACTV, SINK) IH_STEP_NP(LAST, IN, BIAS, ACTV, SINK, 3)
#define IH_STEP_NP(LAST, IN, BIAS, ACTV, SINK, NPR)                                                                                   \
        {                                                                                                                             \
            stream_step_spread_at<IH_CH4, IH_NT>(snext, voff, cur, nxt, [&](const f32x4* w, auto&& piece) {                            \
                dense_blocks_f16x3_hook<8, IH_OB, ACTV, true, false, NPR>(IN, (const half8*)w, BIAS, lane, SINK,                       \
                                                    [&](int ob, int kb) { if (ob == 0 && (kb & 1) == 0) piece(kb >> 1); if (IH_CH4 / IH_NT > 4 && ob == 0 && (kb & 1)) piece(4 + (kb >> 1)); }); }); \
            snext = (LAST) ? wfirst : snext + (size_t)IH_CH4 * 16;                                                                     \
            asm volatile("" : "+s"(snext));      /* opaque: one running pointer, not 18 hoisted ones */                                \
        }
#pragma unroll
        for (int c = 0; c < 16 / IH_OB; ++c)                           // fc2: output blocks IH_OB c .. = k-blocks IH_OB/2 c .. of fc3
            IH_STEP(false, x, bias4 + 4 * IH_OB * c, 1, ([&](int p, const f32x4& o0, const f32x4& o1) { y[IH_OB / 2 * c + p] = split_f16_r(amax, o0, o1); }));
#pragma unroll
        for (int c = 0; c < 16 / IH_OB; ++c)                           // fc3
            IH_STEP(false, y, bias4 + 64 + 4 * IH_OB * c, 1, ([&](int p, const f32x4& o0, const f32x4& o1) { x[IH_OB / 2 * c + p] = split_f16_r(amax, o0, o1); }));
        f32x4 b[4];
#pragma unroll
        for (int c = 0; c < 4 / IH_OB; ++c)                            // fc_query: 64 heads
            IH_STEP_NP(c + 32 / IH_OB + 2 == IH_NCH, x, bias4 + 128 + 4 * IH_OB * c, 0,
                       ([&](int p, const f32x4& o0, const f32x4& o1) { b[IH_OB * c + 2 * p] = o0; b[IH_OB * c + 2 * p + 1] = o1; }), PPS_FCQ_PRODUCTS);
#undef IH_STEP
#undef IH_STEP_NP
        __builtin_amdgcn_s_setprio(0);
        // ---- softmax over the 64 neighbours (4 waves x 16 rows) for each of the 64 heads: as in interp_pool_kernel -------------
        float e[16];
        {
            f32x4 m4[4], s4[4];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                // row maxima and row sums of four heads at a time, the lane permutation folded into v_max_f32_dpp / v_add_f32_dpp (pps_common.h)
                float v[4], mx[4], sm[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) mx[r] = v[r] = valid ? b[bb][r] : -INFINITY;
                row16_max4(mx[0], mx[1], mx[2], mx[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ev = valid ? __expf(v[r] - mx[r]) : 0.f;
                    e[bb * 4 + r] = ev;
                    sm[r] = ev;
                }
                row16_sum4(sm[0], sm[1], sm[2], sm[3]);
                m4[bb] = f32x4{mx[0], mx[1], mx[2], mx[3]};
                s4[bb] = f32x4{sm[0], sm[1], sm[2], sm[3]};
            }
            if (n < 4) {
                const f32x4 mm = (n == 0) ? m4[0] : (n == 1) ? m4[1] : (n == 2) ? m4[2] : m4[3];
                const f32x4 ss = (n == 0) ? s4[0] : (n == 1) ? s4[1] : (n == 2) ? s4[2] : s4[3];
                ((f32x4*)(msm + wave * 64))[4 * n + g] = mm;
                ((f32x4*)(mss + wave * 64))[4 * n + g] = ss;
            }
        }
        __syncthreads();
        {
            float mw[4], sw[4];
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) { mw[w2] = msm[(wbase + w2) * 64 + lane]; sw[w2] = mss[(wbase + w2) * 64 + lane]; }
            const float M = fmaxf(fmaxf(mw[0], mw[1]), fmaxf(mw[2], mw[3]));
            float S = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) S += sw[w2] * __expf(mw[w2] - M);
            f_l[wave * 64 + lane] = __expf(mw[wq] - M) / (64.f * S);
        }
        __syncthreads();
        float an = 0.f;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x4 f4 = ((const f32x4*)(f_l + wave * 64))[4 * bb + g];
#pragma unroll
            for (int r = 0; r < 4; ++r) an += e[bb * 4 + r] * f4[r];
        }
        an += __shfl_xor(an, 16);
        an += __shfl_xor(an, 32);           // attention weight of row n (mean over the 64 heads)
        f32x4 a[16];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {                               // h3 back to fp32 (hi + lo) in the standard block layout
            join_f16(x[kb], a[2 * kb], a[2 * kb + 1]);
            a[2 * kb] *= an;
            a[2 * kb + 1] *= an;
        }
        rows16_sum_transposed(a, lane);
        ((f32x4*)(part + wave * 256))[4 * n + g] = a[0];
        __syncthreads();
        if (wq == 0 && qv) {
            f32x4 s = ((const f32x4*)(part + (wbase + 0) * 256))[lane];
            s += ((const f32x4*)(part + (wbase + 1) * 256))[lane];
            s += ((const f32x4*)(part + (wbase + 2) * 256))[lane];
            s += ((const f32x4*)(part + (wbase + 3) * 256))[lane];
            ((f32x4*)(pooled + qi * 256))[lane] = s;
        }
    }
    range_commit(amax, range);
}

// =====================================================================================================
// interp_small: the POCO projection head (source/poco_model.py:362-419 with latent_size C = 16*CB <= 64 and a handful of
// output channels, configs/poco.yaml:47-48: C = 32, out = 2).  Same register-tile chain as interp_pool; all weights
// (a few KiB) stay resident in LDS, and fc8 . fc_value (composed on the host) is applied to the pooled feature in place.
// weights (floats): [xyz 64*CB][fc2 256*CB*CB][fc3 256*CB*CB][fcq 1024*CB]   bias: [16CB][16CB][64]   tail: [NOUT][16CB] + [NOUT]
// =====================================================================================================
#define IS_NT 256
#define IS_MAX_OUT 8
// H = true: fc2, fc3 and fc_query in split precision on the f16 matrix pipe (pps_common.h); `w16` = f16x3 packs of the three layers (as many bytes
// as their fp32 packs: the LDS layout does not change), `guard[0]` receives the range flag.  H = false with a non-null `guard`: the fp32 fall-back
// behind a split-precision launch -- nothing to do unless guard[0] was raised; guard[1] counts the chunks it recomputed.
template <int CB, bool H>
__global__ __launch_bounds__(IS_NT) void interp_small_kernel(const float* __restrict__ G, const float* __restrict__ pts,
                                                             const float* __restrict__ query, const int64_t* __restrict__ idx,
                                                             int64_t Q, int k, const float* __restrict__ wpack, const float* __restrict__ w16,
                                                             const float* __restrict__ bias, const float* __restrict__ wtail, int nout,
                                                             float* __restrict__ out, int* __restrict__ guard) {
    constexpr int C = 16 * CB, NWX = 64 * CB, NW2 = 256 * CB * CB, NWQ = 1024 * CB, NB = 2 * C + 64, KB = CB / 2;
    if (!H) {
        if (gate_closed(guard)) return;
        if (guard != nullptr && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(guard + 1, 1);
    }
    float amax = 0.f;
    __shared__ __attribute__((aligned(16))) float lds[NWX + 2 * NW2 + NWQ + NB + 4 * 64 * 3 + 4 * C + IS_MAX_OUT * (C + 1)];
    float* xyz_l = lds;
    const f32x4* w2 = (const f32x4*)(lds + NWX);
    const f32x4* w3 = (const f32x4*)(lds + NWX + NW2);
    const f32x4* wq = (const f32x4*)(lds + NWX + 2 * NW2);
    float* bias_l = lds + NWX + 2 * NW2 + NWQ;
    float* msm = bias_l + NB;
    float* mss = msm + 256;
    float* f_l = mss + 256;
    float* part = f_l + 256;                 // [4][C]
    float* tail_l = part + 4 * C;            // [nout][C] then [nout]
    const f32x4* bias4 = (const f32x4*)bias_l;
    const int lane = lane_id(), wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    for (int i = threadIdx.x; i < NWX; i += IS_NT) lds[i] = wpack[i];
    {
        const float* dense = H ? w16 : wpack + NWX;
        for (int i = threadIdx.x; i < 2 * NW2 + NWQ; i += IS_NT) lds[NWX + i] = dense[i];
    }
    for (int i = threadIdx.x; i < NB; i += IS_NT) bias_l[i] = bias[i];
    for (int i = threadIdx.x; i < nout * (C + 1); i += IS_NT) tail_l[i] = wtail[i];
    __syncthreads();
    for (int64_t qi = blockIdx.x; qi < Q; qi += gridDim.x) {
        const int row = wave * 16 + n;
        const bool valid = row < k;
        const int64_t i = idx[qi * k + (valid ? row : 0)];
        f32x4 a[CB], h[CB], b[4];
        const f32x4* grow = (const f32x4*)(G + i * C) + g;
#pragma unroll
        for (int bb = 0; bb < CB; ++bb) a[bb] = grow[4 * bb];
        const float coord = (g < 3) ? (query[qi * 3 + g] - pts[i * 3 + g]) : 0.f;
        xyz_blocks<CB>(coord, a, xyz_l, lane);
        relu_blocks<CB>(a);
        if constexpr (H) {
            HiLo x[KB], y[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) x[kb] = split_f16_r(amax, a[2 * kb], a[2 * kb + 1]);
            dense_blocks_f16x3<KB, CB, 1, false>(x, (const half8*)w2, bias4, lane,
                                                 [&](int p, const f32x4& o0, const f32x4& o1) { y[p] = split_f16_r(amax, o0, o1); });
            dense_blocks_f16x3<KB, CB, 1, false>(y, (const half8*)w3, bias4 + 4 * CB, lane, [&](int p, const f32x4& o0, const f32x4& o1) {
                x[p] = split_f16_r(amax, o0, o1);
                a[2 * p] = o0;                                            // the pooled feature keeps the fp32 output of fc3
                a[2 * p + 1] = o1;
            });
            dense_blocks_f16x3<KB, 4, 0, false>(x, (const half8*)wq, bias4 + 8 * CB, lane,
                                                [&](int p, const f32x4& o0, const f32x4& o1) { b[2 * p] = o0; b[2 * p + 1] = o1; });
        } else {
            dense_blocks<CB, CB, 1>(a, h, w2, bias4, lane);
            dense_blocks<CB, CB, 1>(h, a, w3, bias4 + 4 * CB, lane);
            dense_blocks<CB, 4, 0>(a, b, wq, bias4 + 8 * CB, lane);       // 64 heads
        }
        float e[16];
        {
            f32x4 m4[4], s4[4];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = valid ? b[bb][r] : -INFINITY;
                    const float mx = row16_max(v);
                    const float ev = valid ? __expf(v - mx) : 0.f;
                    e[bb * 4 + r] = ev;
                    m4[bb][r] = mx;
                    s4[bb][r] = row16_sum(ev);
                }
            if (n < 4) {
                const f32x4 mm = (n == 0) ? m4[0] : (n == 1) ? m4[1] : (n == 2) ? m4[2] : m4[3];
                const f32x4 ss = (n == 0) ? s4[0] : (n == 1) ? s4[1] : (n == 2) ? s4[2] : s4[3];
                ((f32x4*)(msm + wave * 64))[4 * n + g] = mm;
                ((f32x4*)(mss + wave * 64))[4 * n + g] = ss;
            }
        }
        __syncthreads();
        {
            float mw[4], sw[4];
#pragma unroll
            for (int w2i = 0; w2i < 4; ++w2i) { mw[w2i] = msm[w2i * 64 + lane]; sw[w2i] = mss[w2i * 64 + lane]; }
            const float M = fmaxf(fmaxf(mw[0], mw[1]), fmaxf(mw[2], mw[3]));
            float S = 0.f;
#pragma unroll
            for (int w2i = 0; w2i < 4; ++w2i) S += sw[w2i] * __expf(mw[w2i] - M);
            f_l[wave * 64 + lane] = __expf(mw[wave] - M) / (64.f * S);
        }
        __syncthreads();
        float an = 0.f;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const f32x4 f4 = ((const f32x4*)(f_l + wave * 64))[4 * bb + g];
#pragma unroll
            for (int r = 0; r < 4; ++r) an += e[bb * 4 + r] * f4[r];
        }
        an += __shfl_xor(an, 16);
        an += __shfl_xor(an, 32);
#pragma unroll
        for (int bb = 0; bb < CB; ++bb) {
            f32x4 p;
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = row16_sum(an * a[bb][r]);
            if (n == 0) ((f32x4*)(part + wave * C))[4 * bb + g] = p;
        }
        __syncthreads();
        if (threadIdx.x < nout) {
            float acc = tail_l[nout * C + threadIdx.x];
            for (int c = 0; c < C; ++c) acc += tail_l[threadIdx.x * C + c] * (part[c] + part[C + c] + part[2 * C + c] + part[3 * C + c]);
            out[qi * nout + threadIdx.x] = acc;
        }
    }
    if (H) range_commit(amax, guard);
}

// =====================================================================================================
// PointNet phase A: conv0a, conv0b, stn.conv1..3 (+ReLU), max over the patch -> g[q,256]
// weights (floats): [xyz 256][c0b 4096][s1 4096][s2 8192][s3 32768]   bias [64][64][64][128][256]
//
// Row packing (both PointNet row kernels): a patch has P = 16*FB + LO rows.  When LO is 2, 4 or 8 (P = 50, 100, 200 of
// configs/ppsurf_*nn.yaml) a wave owns QG = 16/LO queries at a time and evaluates their LO left-over rows TOGETHER in
// one 16-row tile (group-reduced per query and parked in LDS), then the FB full tiles of each query: no padded rows
// (P = 50: 25 tiles per 8 queries instead of 32).  Any other P falls back to ceil(P/16) tiles per query with the
// padding rows repeating a valid point (max) / masked (softmax).
// =====================================================================================================
// Two INDEPENDENT 4-wave workgroups per CU (16 KiB weight chunks so that both fit the LDS next to the parked rows) instead of
// one 8-wave workgroup with 32 KiB chunks: the two waves of a SIMD then belong to different workgroups, drift out of phase and --
// with the wave priority raised for the MFMA phase -- keep the matrix pipe fed while the other one reduces / parks
// (stn_rows 2.00 -> 1.95 ms, feat_rows 2.47 -> 2.30 ms per 50000 queries; PNT=512 / PCH4=2048 is the old configuration).
#ifndef PNT
#define PNT 256                // threads of the PointNet row kernels
#endif
#ifndef PCH4
#define PCH4 1024              // f32x4 per streamed weight chunk of the PointNet row kernels (1024 = 16 KiB)
#endif
#define PNW (PNT / 64)
#define PN_WG_PER_CU (512 / PNT)
#define PN_C2N (2048 / PCH4)   // chunks of conv2 (64 -> 128: 2048 f32x4) and of conv3 (128 -> 256: 8192 f32x4)
#define PN_C3N (8192 / PCH4)
#define PN_C2OB (8 / PN_C2N)   // output blocks per chunk
#define PN_C3OB (16 / PN_C3N)
#define PN_ROWF 260            // floats per parked left-over row (256 + pad against bank conflicts)
#define PA_W_XYZ 256
#define PA_NBIAS 576
#define PN_PARK_ROWS 8         // parked rows per wave = max queries per wave group
#define PA_LDS_BYTES (2 * PCH4 * 16 + (PA_W_XYZ + PA_NBIAS + PNW * PN_PARK_ROWS * PN_ROWF) * 4)

struct PatchPacking {
    int fb, lo, qg, tiles_per_query;      // full tiles, left-over rows, queries per wave group, tiles evaluated per query
    bool packed;
};
// mode 0: padded tiles; 1: packed, QG = 16/LO queries per wave group; 2: the packed ARITHMETIC with one query per group (its LO left-over
// rows alone in a tile, the other columns idle).  Modes 1 and 2 give bit-identical results per query -- the group reductions run over the
// same aligned LO lanes, the per-query transform product only ever adds exact zeros for the other columns -- so the launch may use full
// packed rounds for most queries and mode 2 for the remainder, and a query's logits do not depend on where in a chunk it sits or on how a
// query list was cut into chunks / sharded over ranks (tests/test_gpu_decoder.py::test_logits_do_not_depend_on_the_chunking).
__host__ __device__ inline PatchPacking patch_packing(int P, int mode = 1) {
    PatchPacking k;
    const int lo = P & 15;
    k.packed = mode != 0 && (P >= 16) && (lo == 2 || lo == 4 || lo == 8);
    k.fb = k.packed ? P / 16 : (P + 15) / 16;
    k.lo = k.packed ? lo : 0;
    k.qg = k.packed ? (mode == 2 ? 1 : 16 / lo) : 1;
    k.tiles_per_query = k.fb;
    return k;
}

// conv0a .. stn.conv3 on one 16-row tile; z = 256 channels
__device__ __forceinline__ void stn_chain(float coord, f32x4 (&z)[16], const float* xyz_l, const f32x4* bias4, const f32x4* wg,
                                          f32x4*& cur, f32x4*& nxt, int lane) {
    const int g = lane >> 4;
    f32x4 x0[4], x1[4], y[8];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) x0[bb] = bias4[4 * bb + g];
    xyz_blocks<4>(coord, x0, xyz_l, lane);
    relu_blocks<4>(x0);
    __builtin_amdgcn_s_setprio(PPS_PRIO_PN);
    stream_step<1024, PNT>(wg + 1024, cur, nxt, [&](const f32x4* w) { dense_blocks<4, 4, 1>(x0, x1, w, bias4 + 16, lane); });
    stream_step<PCH4, PNT>(wg + 2048, cur, nxt, [&](const f32x4* w) { dense_blocks<4, 4, 1>(x1, x0, w, bias4 + 32, lane); });
#pragma unroll
    for (int h = 0; h < PN_C2N; ++h)
        stream_step<PCH4, PNT>(wg + 2048 + (h + 1) * PCH4, cur, nxt,
                               [&](const f32x4* w) { dense_blocks<4, PN_C2OB, 1>(x0, &y[PN_C2OB * h], w, bias4 + 48 + 4 * PN_C2OB * h, lane); });
#pragma unroll
    for (int c = 0; c < PN_C3N - 1; ++c)
        stream_step<PCH4, PNT>(wg + 4096 + (c + 1) * PCH4, cur, nxt,
                               [&](const f32x4* w) { dense_blocks<8, PN_C3OB, 1>(y, &z[PN_C3OB * c], w, bias4 + 80 + 4 * PN_C3OB * c, lane); });
    stream_step<1024, PNT>(wg, cur, nxt, [&](const f32x4* w) {
        dense_blocks<8, PN_C3OB, 1>(y, &z[PN_C3OB * (PN_C3N - 1)], w, bias4 + 80 + 4 * PN_C3OB * (PN_C3N - 1), lane); });
    __builtin_amdgcn_s_setprio(0);
}

// the same chain in split precision (decoder dtype "f16x3"): conv0a stays an fp32 MFMA (K = 3), the four dense layers run as
// three f16 products each; `wg` -> pps_pack_dense_f16x3 images of c0b, s1, s2, s3 (same byte sizes and chunk boundaries as fp32)
__device__ __forceinline__ void stn_chain_h(float coord, f32x4 (&z)[16], const float* xyz_l, const f32x4* bias4, const f32x4* wg,
                                            f32x4*& cur, f32x4*& nxt, int lane, float& amax) {
    const int g = lane >> 4;
    HiLo a[2], b[2], y[4];
    {
        f32x4 x0[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) x0[bb] = bias4[4 * bb + g];
        xyz_blocks<4>(coord, x0, xyz_l, lane);
        relu_blocks<4>(x0);
        a[0] = split_f16_r(amax, x0[0], x0[1]);
        a[1] = split_f16_r(amax, x0[2], x0[3]);
    }
    __builtin_amdgcn_s_setprio(PPS_PRIO_PN);
    stream_step<1024, PNT>(wg + 1024, cur, nxt, [&](const f32x4* w) {
        dense_blocks_f16x3<2, 4, 1>(a, (const half8*)w, bias4 + 16, lane, [&](int i, const f32x4& o0, const f32x4& o1) { b[i] = split_f16_r(amax, o0, o1); }); });
    stream_step<PCH4, PNT>(wg + 2048, cur, nxt, [&](const f32x4* w) {
        dense_blocks_f16x3<2, 4, 1>(b, (const half8*)w, bias4 + 32, lane, [&](int i, const f32x4& o0, const f32x4& o1) { a[i] = split_f16_r(amax, o0, o1); }); });
#pragma unroll
    for (int h = 0; h < PN_C2N; ++h)
        stream_step<PCH4, PNT>(wg + 2048 + (h + 1) * PCH4, cur, nxt, [&](const f32x4* w) {
            dense_blocks_f16x3<2, PN_C2OB, 1>(a, (const half8*)w, bias4 + 48 + 4 * PN_C2OB * h, lane,
                                              [&](int i, const f32x4& o0, const f32x4& o1) { y[PN_C2OB / 2 * h + i] = split_f16_r(amax, o0, o1); }); });
#pragma unroll
    for (int c = 0; c < PN_C3N - 1; ++c)
        stream_step<PCH4, PNT>(wg + 4096 + (c + 1) * PCH4, cur, nxt, [&](const f32x4* w) {
            dense_blocks_f16x3<4, PN_C3OB, 1>(y, (const half8*)w, bias4 + 80 + 4 * PN_C3OB * c, lane,
                                              [&](int i, const f32x4& o0, const f32x4& o1) { z[PN_C3OB * c + 2 * i] = o0; z[PN_C3OB * c + 2 * i + 1] = o1; }); });
    stream_step<1024, PNT>(wg, cur, nxt, [&](const f32x4* w) {
        dense_blocks_f16x3<4, PN_C3OB, 1>(y, (const half8*)w, bias4 + 80 + 4 * PN_C3OB * (PN_C3N - 1), lane,
                                          [&](int i, const f32x4& o0, const f32x4& o1) { z[PN_C3OB * (PN_C3N - 1) + 2 * i] = o0; z[PN_C3OB * (PN_C3N - 1) + 2 * i + 1] = o1; }); });
    __builtin_amdgcn_s_setprio(0);
}

// The split-precision chain for T tiles of one wave at once (pps_common.h, dense_blocks_f16x3_tiles): one pass over the streamed weights and one set
// of A-fragment reads serve T x 16 rows.  conv3's output blocks are handed to zsink(tile, first_block, o0, o1) as they leave the matrix pipe (the
// callers reduce them at once: a tile's 256 channels are never held).  Per tile the arithmetic is exactly that of stn_chain_h: results do not
// depend on how tiles are grouped.
template <int T, class ZSink>
__device__ __forceinline__ void stn_chain_h_tiles(const float (&coord)[T], const float* xyz_l, const f32x4* bias4, const f32x4* wg,
                                                  f32x4*& cur, f32x4*& nxt, int lane, float& amax, ZSink&& zsink) {
    const int g = lane >> 4;
    HiLo a[T][2], b[T][2], y[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        f32x4 x0[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) x0[bb] = bias4[4 * bb + g];
        xyz_blocks<4>(coord[t], x0, xyz_l, lane);
        relu_blocks<4>(x0);
        a[t][0] = split_f16_r(amax, x0[0], x0[1]);
        a[t][1] = split_f16_r(amax, x0[2], x0[3]);
    }
    __builtin_amdgcn_s_setprio(PPS_PRIO_PN);
    stream_step<1024, PNT>(wg + 1024, cur, nxt, [&](const f32x4* w) {
        dense_blocks_f16x3_tiles<T, 2, 4, 1>(a, (const half8*)w, bias4 + 16, lane, [&](int t, int i, const f32x4& o0, const f32x4& o1) { b[t][i] = split_f16_r(amax, o0, o1); }); });
    stream_step<PCH4, PNT>(wg + 2048, cur, nxt, [&](const f32x4* w) {
        dense_blocks_f16x3_tiles<T, 2, 4, 1>(b, (const half8*)w, bias4 + 32, lane, [&](int t, int i, const f32x4& o0, const f32x4& o1) { a[t][i] = split_f16_r(amax, o0, o1); }); });
#pragma unroll
    for (int h = 0; h < PN_C2N; ++h)
        stream_step<PCH4, PNT>(wg + 2048 + (h + 1) * PCH4, cur, nxt, [&](const f32x4* w) {
            dense_blocks_f16x3_tiles<T, 2, PN_C2OB, 1>(a, (const half8*)w, bias4 + 48 + 4 * PN_C2OB * h, lane,
                                                      [&](int t, int i, const f32x4& o0, const f32x4& o1) { y[t][PN_C2OB / 2 * h + i] = split_f16_r(amax, o0, o1); }); });
#pragma unroll
    for (int c = 0; c < PN_C3N - 1; ++c)
        stream_step<PCH4, PNT>(wg + 4096 + (c + 1) * PCH4, cur, nxt, [&](const f32x4* w) {
            dense_blocks_f16x3_tiles<T, 4, PN_C3OB, 1>(y, (const half8*)w, bias4 + 80 + 4 * PN_C3OB * c, lane,
                                                      [&](int t, int i, const f32x4& o0, const f32x4& o1) { zsink(t, PN_C3OB * c + 2 * i, o0, o1); }); });
    stream_step<1024, PNT>(wg, cur, nxt, [&](const f32x4* w) {
        dense_blocks_f16x3_tiles<T, 4, PN_C3OB, 1>(y, (const half8*)w, bias4 + 80 + 4 * PN_C3OB * (PN_C3N - 1), lane,
                                                  [&](int t, int i, const f32x4& o0, const f32x4& o1) { zsink(t, PN_C3OB * (PN_C3N - 1) + 2 * i, o0, o1); }); });
    __builtin_amdgcn_s_setprio(0);
}

// H = false: fp32 (wdense = wpack + 256 floats of the same image); H = true: split precision (wdense = the f16x3 image)
template <bool H>
__global__ __launch_bounds__(PNT, 2) void pointnet_stn_rows_kernel(const float* __restrict__ patches, int64_t Q, int P, int pack,
                                                                   const float* __restrict__ wpack, const f32x4* __restrict__ wdense,
                                                                   const float* __restrict__ bias, float* __restrict__ gout, int* __restrict__ flag) {
    // flag: H -- where the range guard of the split reports (pps_common.h); !H -- gate of the fp32 fallback (may be null)
    if (!H && gate_closed(flag)) return;
    float amax = 0.f;
    f32x4* buf0 = (f32x4*)pps_smem;
    f32x4* buf1 = buf0 + PCH4;
    float* xyz_l = (float*)(buf1 + PCH4);
    float* bias_l = xyz_l + PA_W_XYZ;
    float* park = bias_l + PA_NBIAS;                       // [PNW][PN_PARK_ROWS][PN_ROWF] per-query left-over maxima
    const f32x4* bias4 = (const f32x4*)bias_l;
    const f32x4* wg = wdense;
    const int lane = lane_id(), wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    float* mypark = park + wave * PN_PARK_ROWS * PN_ROWF;

    lds_fill(xyz_l, wpack, PA_W_XYZ);
    lds_fill(bias_l, bias, PA_NBIAS);
    stream_prologue<1024, PNT>(wg, buf0);
    stream_wait();
    __syncthreads();
    f32x4 *cur = buf0, *nxt = buf1;

    const PatchPacking pk = patch_packing(P, pack);
    const int64_t ngroups = (Q + pk.qg - 1) / pk.qg;
    const int ntiles = (int)((ngroups + PNW - 1) / PNW);
    int first, count, stride;
    xcd_tile_range(ntiles, first, count, stride);
    for (int it = 0; it < count; ++it) {
        const int64_t q0 = ((int64_t)(first + it * stride) * PNW + wave) * pk.qg;
        if constexpr (H) {
            // split precision: conv3's blocks are reduced as they are produced, and the full tiles of a query go through the chain two at a time
            if (pk.packed) {
                const int ql = n / pk.lo;
                const int64_t qq = (q0 + ql < Q) ? q0 + ql : Q - 1;
                const float coord[1] = {(g < 3) ? patches[(qq * P + pk.fb * 16 + (n % pk.lo)) * 3 + g] : 0.f};
                stn_chain_h_tiles<1>(coord, xyz_l, bias4, wg, cur, nxt, lane, amax, [&](int, int bb, const f32x4& o0, const f32x4& o1) {
                    f32x4 m0, m1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { m0[r] = group_max(o0[r], pk.lo); m1[r] = group_max(o1[r], pk.lo); }
                    if ((n % pk.lo) == 0) { ((f32x4*)(mypark + ql * PN_ROWF))[4 * bb + g] = m0; ((f32x4*)(mypark + ql * PN_ROWF))[4 * (bb + 1) + g] = m1; }
                });
            }
            for (int qi = 0; qi < pk.qg; ++qi) {
                const int64_t q = q0 + qi;
                const bool qv = q < Q;
                const int64_t qc = qv ? q : Q - 1;
                f32x4 rmax[16];
#pragma unroll
                for (int bb = 0; bb < 16; ++bb)
                    rmax[bb] = pk.packed ? ((const f32x4*)(mypark + qi * PN_ROWF))[4 * bb + g] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                auto zmax = [&](int, int bb, const f32x4& o0, const f32x4& o1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { rmax[bb][r] = max_raw(rmax[bb][r], o0[r]); rmax[bb + 1][r] = max_raw(rmax[bb + 1][r], o1[r]); }
                };
                auto row_coord = [&](int rb) {
                    const int row = rb * 16 + n;
                    const int rowc = row < P ? row : P - 1;       // padded rows repeat a valid point: max unaffected
                    return (g < 3) ? patches[(qc * P + rowc) * 3 + g] : 0.f;
                };
                int rb = 0;
                for (; rb + 1 < pk.fb; rb += 2) {
                    const float coord[2] = {row_coord(rb), row_coord(rb + 1)};
                    stn_chain_h_tiles<2>(coord, xyz_l, bias4, wg, cur, nxt, lane, amax, zmax);
                }
                if (rb < pk.fb) {
                    const float coord[1] = {row_coord(rb)};
                    stn_chain_h_tiles<1>(coord, xyz_l, bias4, wg, cur, nxt, lane, amax, zmax);
                }
#pragma unroll
                for (int bb = 0; bb < 16; ++bb) {
                    float p0 = rmax[bb][0], p1 = rmax[bb][1], p2 = rmax[bb][2], p3 = rmax[bb][3];
                    row16_max4(p0, p1, p2, p3);                   // (the lane permutation folded into v_max_f32_dpp, pps_common.h)
                    if (n == 0 && qv) ((f32x4*)(gout + q * 256))[4 * bb + g] = f32x4{p0, p1, p2, p3};
                }
            }
            continue;
        }
        f32x4 z[16];
        if (pk.packed) {
            // the LO left-over rows of the QG queries of this wave, one tile
            const int ql = n / pk.lo;
            const int64_t qq = (q0 + ql < Q) ? q0 + ql : Q - 1;
            const float coord = (g < 3) ? patches[(qq * P + pk.fb * 16 + (n % pk.lo)) * 3 + g] : 0.f;
            if (H) stn_chain_h(coord, z, xyz_l, bias4, wg, cur, nxt, lane, amax); else stn_chain(coord, z, xyz_l, bias4, wg, cur, nxt, lane);
#pragma unroll
            for (int bb = 0; bb < 16; ++bb) {
                f32x4 m;
#pragma unroll
                for (int r = 0; r < 4; ++r) m[r] = group_max(z[bb][r], pk.lo);
                if ((n % pk.lo) == 0) ((f32x4*)(mypark + ql * PN_ROWF))[4 * bb + g] = m;
            }
        }
        for (int qi = 0; qi < pk.qg; ++qi) {
            const int64_t q = q0 + qi;
            const bool qv = q < Q;
            const int64_t qc = qv ? q : Q - 1;
            f32x4 rmax[16];
#pragma unroll
            for (int bb = 0; bb < 16; ++bb)
                rmax[bb] = pk.packed ? ((const f32x4*)(mypark + qi * PN_ROWF))[4 * bb + g] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (int rb = 0; rb < pk.fb; ++rb) {
                const int row = rb * 16 + n;
                const int rowc = row < P ? row : P - 1;       // padded rows repeat a valid point: max unaffected
                const float coord = (g < 3) ? patches[(qc * P + rowc) * 3 + g] : 0.f;
                if (H) stn_chain_h(coord, z, xyz_l, bias4, wg, cur, nxt, lane, amax); else stn_chain(coord, z, xyz_l, bias4, wg, cur, nxt, lane);
#pragma unroll
                for (int bb = 0; bb < 16; ++bb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rmax[bb][r] = max_raw(rmax[bb][r], z[bb][r]);
            }
#pragma unroll
            for (int bb = 0; bb < 16; ++bb) {
                float p0 = rmax[bb][0], p1 = rmax[bb][1], p2 = rmax[bb][2], p3 = rmax[bb][3];
                row16_max4(p0, p1, p2, p3);
                const f32x4 p = {p0, p1, p2, p3};
                if (n == 0 && qv) ((f32x4*)(gout + q * 256))[4 * bb + g] = p;
            }
        }
    }
    if (H) range_commit(amax, flag);
}

// =====================================================================================================
// PointNet phase B: g[q,256] -> fc1(128, ReLU) -> fc2(64, ReLU) -> fc3(4096) (+I in bias) -> trans2[q,4096]
// weights (floats): [fc1 32768][fc2 8192][fc3 262144]   bias [128][64][4096]
// =====================================================================================================
#define PB_NBIAS (128 + 64 + 4096)
// LDS: the two 32 KiB weight buffers + the 16 KiB bias of the last layer = exactly 80 KiB, so that TWO workgroups fit a CU's 160 KiB.  (Rounds 1-3
// also kept the 768 bytes of the first two layers' biases there: 165 376 bytes for two workgroups, 1.5 KiB too many -- the kernel ran ONE 4-wave
// workgroup per CU.)  The small biases are read from global memory, five times per tile.
#define PB_LDS_BYTES (2 * CH4 * 16 + 4096 * 4)
#ifndef PB_T
#define PB_T 1                 // 16-query tiles a wave carries through fc3 together (2: no gain, 3: slower -- measured)
#endif

__global__ __launch_bounds__(NT, 2) void pointnet_stn_fc_kernel(const float* __restrict__ gin, int64_t Q,
                                                                const float* __restrict__ wpack, const float* __restrict__ bias,
                                                                float* __restrict__ trans2, const int* __restrict__ gate) {
    if (gate_closed(gate)) return;
    f32x4* buf0 = (f32x4*)pps_smem;
    f32x4* buf1 = buf0 + CH4;
    float* bias_l = (float*)(buf1 + CH4);
    const f32x4* bias4 = (const f32x4*)bias_l;
    const f32x4* wg = (const f32x4*)wpack;
    const int lane = lane_id(), wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;

    lds_fill(bias_l, bias + 192, 4096);
    const f32x4* bias_g = (const f32x4*)bias;                  // biases of the first two layers: [128][64] floats
    stream_prologue<CH4>(wg, buf0);
    stream_wait();
    __syncthreads();
    f32x4 *cur = buf0, *nxt = buf1;

    // A wave carries PB_T tiles of 16 queries through the last layer together: its 1 MiB of weights is the bulk of the stream,
    // and with one tile per wave the two workgroups of a CU would need 16 B/clk of LDS-DMA to keep the MFMA pipe busy (9.8 TB/s
    // chip-wide, above what the path delivers); fc1 / fc2 (0.16 MiB) are simply streamed once per tile.
    const int ntiles = (int)((Q + NW * 16 * PB_T - 1) / (NW * 16 * PB_T));
    int first, count, stride;
    xcd_tile_range(ntiles, first, count, stride);
    for (int it = 0; it < count; ++it) {
        f32x4 u[PB_T][4];
        int64_t qcs[PB_T];
        bool qvs[PB_T];
#pragma unroll
        for (int t = 0; t < PB_T; ++t) {
            const int64_t qi = (int64_t)(first + it * stride) * (NW * 16 * PB_T) + (wave * PB_T + t) * 16 + n;
            qvs[t] = qi < Q;
            qcs[t] = qvs[t] ? qi : Q - 1;
            f32x4 a[16], h[8];
            {
                const f32x4* src = (const f32x4*)(gin + qcs[t] * 256) + g;
#pragma unroll
                for (int bb = 0; bb < 16; ++bb) a[bb] = src[4 * bb];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                stream_step<CH4>(wg + (c + 1) * CH4, cur, nxt,
                               [&](const f32x4* w) { dense_blocks<16, 2, 1>(a, &h[2 * c], w, bias_g + 8 * c, lane); });
            stream_step<CH4>(t + 1 < PB_T ? wg : wg + 5 * CH4, cur, nxt,
                           [&](const f32x4* w) { dense_blocks<8, 4, 1>(h, u[t], w, bias_g + 32, lane); });
        }
#pragma unroll 1
        for (int c = 0; c < 32; ++c) {
            f32x4 o[PB_T][8];
            const f32x4* gn = (c + 1 < 32) ? wg + (6 + c) * CH4 : wg;
            stream_step<CH4>(gn, cur, nxt, [&](const f32x4* w) {
#pragma unroll
                for (int t = 0; t < PB_T; ++t) dense_blocks<4, 8, 0>(u[t], o[t], w, bias4 + 32 * c, lane);
            });
#pragma unroll
            for (int t = 0; t < PB_T; ++t) {
                if (qvs[t]) {
                    f32x4* dst = (f32x4*)(trans2 + qcs[t] * 4096) + g;
#pragma unroll
                    for (int j = 0; j < 8; ++j) dst[4 * (8 * c + j)] = o[t][j];
                }
            }
        }
    }
}

// Split-precision variant ("f16x3"): the three layers as f16 hi/lo products, and trans2 is written ALREADY SPLIT in the A-operand
// fragment order of pointnet_feat_rows_kernel<true>:  trans2h[q][ob 4][kb 2][part 2][slot 4 m + g][8 halfs]  (the four k-groups of a row
// are adjacent, so the four lanes (query n, g = 0..3) of the writer store 64 contiguous bytes, like the fp32 kernel does)  with
// element j of lane (m, g) = trans2[q][16 ob + m][32 kb + 16 (j >> 2) + 4 g + (j & 3)]  (16 KiB per query, like the fp32 matrix):
// the pair of output blocks (4a + 2kb, 4a + 2kb + 1) of fc3 held by lane (query n, g) IS that fragment for row a = 16 ob + m.
__global__ __launch_bounds__(NT, 2) void pointnet_stn_fc_h_kernel(const float* __restrict__ gin, int64_t Q, const f32x4* __restrict__ w16,
                                                                  const float* __restrict__ bias, half8* __restrict__ trans2h, int* __restrict__ range) {
    float amax = 0.f;
    f32x4* buf0 = (f32x4*)pps_smem;
    f32x4* buf1 = buf0 + CH4;
    float* bias_l = (float*)(buf1 + CH4);
    const f32x4* bias4 = (const f32x4*)bias_l;
    const f32x4* wg = w16;
    const int lane = lane_id(), wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;

    lds_fill(bias_l, bias + 192, 4096);
    const f32x4* bias_g = (const f32x4*)bias;                  // biases of the first two layers: [128][64] floats
    stream_prologue<CH4>(wg, buf0);
    stream_wait();
    __syncthreads();
    f32x4 *cur = buf0, *nxt = buf1;
    const int ntiles = (int)((Q + NW * 16 - 1) / (NW * 16));
    int first, count, stride;
    xcd_tile_range(ntiles, first, count, stride);
    for (int it = 0; it < count; ++it) {
        const int64_t qi = (int64_t)(first + it * stride) * (NW * 16) + wave * 16 + n;
        const bool qv = qi < Q;
        const int64_t qc = qv ? qi : Q - 1;
        HiLo ah[8], h[4], u[2];
        {
            f32x4 a[16];
            const f32x4* src = (const f32x4*)(gin + qc * 256) + g;
#pragma unroll
            for (int bb = 0; bb < 16; ++bb) a[bb] = src[4 * bb];
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) ah[kb] = split_f16_r(amax, a[2 * kb], a[2 * kb + 1]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            stream_step<CH4>(wg + (c + 1) * CH4, cur, nxt, [&](const f32x4* w) {
                dense_blocks_f16x3<8, 2, 1>(ah, (const half8*)w, bias_g + 8 * c, lane, [&](int, const f32x4& o0, const f32x4& o1) { h[c] = split_f16_r(amax, o0, o1); }); });
        stream_step<CH4>(wg + 5 * CH4, cur, nxt, [&](const f32x4* w) {
            dense_blocks_f16x3<4, 4, 1>(h, (const half8*)w, bias_g + 32, lane, [&](int i, const f32x4& o0, const f32x4& o1) { u[i] = split_f16_r(amax, o0, o1); }); });
        half8* dst = trans2h + qc * 1024 + g;                         // + ((ob*2 + kb)*2 + part)*64 + 4*m
#pragma unroll 1
        for (int c = 0; c < 32; ++c) {
            const f32x4* gn = (c + 1 < 32) ? wg + (6 + c) * CH4 : wg;
            stream_step<CH4>(gn, cur, nxt, [&](const f32x4* w) {
                dense_blocks_f16x3<2, 8, 0, false>(u, (const half8*)w, bias4 + 32 * c, lane, [&](int i, const f32x4& o0, const f32x4& o1) {
                    // blocks 8c + 2i, 8c + 2i + 1: row a = 2c + (i >> 1), k-block kb = i & 1
                    const int a = 2 * c + (i >> 1), kb = i & 1;
                    const HiLo v = split_f16_r(amax, o0, o1);
                    if (qv) {
                        half8* d = dst + (((a >> 4) * 2 + kb) * 2) * 64 + 4 * (a & 15);
                        d[0] = v.hi;
                        d[64] = v.lo;
                    }
                }); });
        }
    }
    range_commit(amax, range);
}

// =====================================================================================================
// PointNet phase C: conv0a, conv0b, x <- M x (M = conv1 . trans2 of phase B) + conv1's bias, ReLU, conv2 (+ReLU), attention pooling of conv2's output
// weights (floats): [xyz 256][c0b 4096][c2 8192]   bias [64][64][c1 64][128][256 unused][u = W3^T wq 128 | pad 128][wq.b3 + bq, 0, 0, 0]
// output xbar [q,128]: the attention-pooled conv2 features (conv1 lives in the per-query matrix, conv3 in the tail: decoder.py)
//
// One wave = one query at a time.  With conv1 and conv3 composed away the layer weights are 48 KiB (both dtypes) and stay RESIDENT in LDS: no
// weight stream, no barrier in the main loop, the waves of a workgroup are independent.  What bounds the kernel is the per-query matrix M (16 KiB per
// query, written by phase B): a wave loads it ONCE into registers (64 VGPRs of A fragments) and applies it to all ceil(P/16) tiles of its query --
// the round-3 kernel fetched it once per tile and once more for the packed left-over rows (4 x 16 KiB per query at P = 50).  Left-over rows are a
// padded tile here (rows >= P repeat a valid point and get weight 0): 4 instead of 3.125 tiles per query at P = 50, 96 MFMAs each -- a quarter of the
// matrix work the kernel did before conv3 moved, for a quarter of the fetch.
// =====================================================================================================
#define PC_W_XYZ 256
#define PC_NBIAS (576 + 256 + 4)
#define PC_WFLOATS (4096 + 8192)       // conv0b + conv2 (fp32 floats; the f16x3 image has the same byte size)
#define PC_LDS_BYTES ((PC_WFLOATS + PC_W_XYZ + PC_NBIAS) * 4)

template <bool H>
__global__ __launch_bounds__(PNT, 2) void pointnet_feat_rows_kernel(const float* __restrict__ patches, const float* __restrict__ trans2,
                                                                    int64_t Q, int P, const float* __restrict__ wpack,
                                                                    const f32x4* __restrict__ wdense, const float* __restrict__ bias,
                                                                    float* __restrict__ xbar, int* __restrict__ flag) {
    if (!H && gate_closed(flag)) return;
    float amax = 0.f;
    f32x4* w_l = (f32x4*)pps_smem;                       // [conv0b 1024 f32x4][conv2 2048 f32x4], packed A fragments
    float* xyz_l = (float*)(w_l + PC_WFLOATS / 4);
    float* bias_l = xyz_l + PC_W_XYZ;
    const f32x4* bias4 = (const f32x4*)bias_l;
    const f32x4* u4 = bias4 + 144;                       // W3^T wq: 128 values in the block layout of conv2's output
    const int lane = lane_id(), wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    for (int i = threadIdx.x; i < PC_WFLOATS / 4; i += PNT) w_l[i] = wdense[i];
    lds_fill(xyz_l, wpack, PC_W_XYZ);
    lds_fill(bias_l, bias, PC_NBIAS);
    __syncthreads();
    const float s0 = bias_l[576 + 256];                   // wq . b3 + bq
    const f32x4* w_c0b = w_l;
    const f32x4* w_c2 = w_l + 1024;
    const int ntile_rows = (P + 15) / 16;

    const int ntiles = (int)((Q + PNW - 1) / PNW);
    int first, count, stride;
    xcd_tile_range(ntiles, first, count, stride);
    for (int it = 0; it < count; ++it) {
        const int64_t q = (int64_t)(first + it * stride) * PNW + wave;
        if (q >= Q) continue;                            // no barrier below: a wave without a query just moves on
        // the per-query matrix as A fragments, once
        half8 th[4][2], tl[4][2];
        f32x4 tf[4][4];
        if (H) {
            const half8* tq = (const half8*)trans2 + q * 1024 + 4 * n + g;       // slot 4 m + g of each 1 KiB fragment block (pointnet_stn_fc_h_kernel)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) { th[ob][kb] = tq[((ob * 2 + kb) * 2) * 64]; tl[ob][kb] = tq[((ob * 2 + kb) * 2 + 1) * 64]; }
        } else {
            const f32x4* tq = (const f32x4*)(trans2 + q * 4096);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) tf[ob][kb] = tq[(16 * ob + n) * 16 + 4 * kb + g];
        }
        // per-lane (unreduced) online-softmax state over the patch points (nn.py:91-93)
        f32x4 acc[8];
        float mrun = -INFINITY, ssum = 0.f;
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) acc[bb] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int rb = 0; rb < ntile_rows; ++rb) {
            const int rowi = rb * 16 + n;
            const bool valid = rowi < P;
            const int rowc = valid ? rowi : P - 1;
            const float coord = (g < 3) ? patches[(q * P + rowc) * 3 + g] : 0.f;
            f32x4 x0[4], y[8];
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) x0[bb] = bias4[4 * bb + g];
            xyz_blocks<4>(coord, x0, xyz_l, lane);
            relu_blocks<4>(x0);
            __builtin_amdgcn_s_setprio(PPS_PRIO_PN);
            float s = 0.f;                                   // attention logit of row n: linear in conv2's output (u = W3^T wq packed by the host)
            if (H) {
                HiLo a[2], b[2];
                a[0] = split_f16_r(amax, x0[0], x0[1]);
                a[1] = split_f16_r(amax, x0[2], x0[3]);
                dense_blocks_f16x3<2, 4, 1>(a, (const half8*)w_c0b, bias4 + 16, lane, [&](int i, const f32x4& o0, const f32x4& o1) { b[i] = split_f16_r(amax, o0, o1); });
                // x <- M x: three f16 products per (output block, k-block), A operands from registers
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    f32x4 o = bias4[32 + 4 * ob + g], c = {0.f, 0.f, 0.f, 0.f};           // conv1's bias seeds the sum
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(th[ob][kb], b[kb].hi, o, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(th[ob][kb], b[kb].lo, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(tl[ob][kb], b[kb].hi, c, 0, 0, 0);
                    }
                    o += c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) x0[ob][r] = fmaxf(o[r], 0.f);
                }
                a[0] = split_f16_r(amax, x0[0], x0[1]);
                a[1] = split_f16_r(amax, x0[2], x0[3]);
                dense_blocks_f16x3<2, 8, 1>(a, (const half8*)w_c2, bias4 + 48, lane, [&](int i, const f32x4& o0, const f32x4& o1) {
                    const f32x4 w0 = u4[4 * (2 * i) + g], w1 = u4[4 * (2 * i + 1) + g];
#pragma unroll
                    for (int r = 0; r < 4; ++r) s += w0[r] * o0[r] + w1[r] * o1[r];
                    y[2 * i] = o0;
                    y[2 * i + 1] = o1;
                });
            } else {
                f32x4 x1[4];
                dense_blocks<4, 4, 1>(x0, x1, w_c0b, bias4 + 16, lane);
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    f32x4 o = bias4[32 + 4 * ob + g];
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        o = __builtin_amdgcn_mfma_f32_16x16x4f32(tf[ob][kb].x, x1[kb].x, o, 0, 0, 0);
                        o = __builtin_amdgcn_mfma_f32_16x16x4f32(tf[ob][kb].y, x1[kb].y, o, 0, 0, 0);
                        o = __builtin_amdgcn_mfma_f32_16x16x4f32(tf[ob][kb].z, x1[kb].z, o, 0, 0, 0);
                        o = __builtin_amdgcn_mfma_f32_16x16x4f32(tf[ob][kb].w, x1[kb].w, o, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) x0[ob][r] = fmaxf(o[r], 0.f);
                }
                dense_blocks<4, 8, 1>(x0, y, w_c2, bias4 + 48, lane);
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) {
                    const f32x4 w4 = u4[4 * bb + g];
#pragma unroll
                    for (int r = 0; r < 4; ++r) s += w4[r] * y[bb][r];
                }
            }
            __builtin_amdgcn_s_setprio(0);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            s += s0;
            // online softmax: advance the state with this tile's logits, then add its rows (conv3 acts on the pooled vector, in the tail)
            const float mblk = row16_max(valid ? s : -INFINITY);
            const float mnew = fmaxf(mrun, mblk);
            const float scale = __expf(mrun - mnew);
            const float en = valid ? __expf(s - mnew) : 0.f;
            mrun = mnew;
            ssum = ssum * scale + en;
#pragma unroll
            for (int bb = 0; bb < 8; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[bb][r] = acc[bb][r] * scale + en * y[bb][r];
        }
        const float inv = 1.f / row16_sum(ssum);
        rows16_sum_transposed8(acc, lane);            // lane (n,g): acc[0] = sum over the 16 lanes of block n & 7
        if (n < 8) ((f32x4*)(xbar + q * 128))[4 * n + g] = acc[0] * inv;
    }
    if (H) range_commit(amax, flag);
}

// =====================================================================================================
// Tail: [pooled 256 | xbar 128] -> 256 (ReLU) -> 256 (ReLU) -> 2      (Wb carries conv3 of the PointNet branch, decoder.py)
// weights (floats): [Wa 65536][Wb 32768][L2 65536][L3 8192]   bias [256][256][32]
// =====================================================================================================
#define TL_NBIAS (256 + 256 + 32)
#define TL_LDS_BYTES (2 * CH4 * 16 + TL_NBIAS * 4)

__global__ __launch_bounds__(NT, 2) void decode_tail_kernel(const float* __restrict__ pooled, const float* __restrict__ xbar, int64_t Q,
                                                            const float* __restrict__ wpack, const float* __restrict__ bias,
                                                            float* __restrict__ logits, float* __restrict__ occ, int* __restrict__ gate) {
    // gate (may be null): [0] = a split-precision kernel of this chunk left the f16 range -> this launch recomputes the chunk in fp32;
    // [1] counts such chunks (read by DecoderPlan.range_fallbacks())
    if (gate_closed(gate)) return;
    if (gate != nullptr && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(gate + 1, 1);
    f32x4* buf0 = (f32x4*)pps_smem;
    f32x4* buf1 = buf0 + CH4;
    float* bias_l = (float*)(buf1 + CH4);
    const f32x4* bias4 = (const f32x4*)bias_l;
    const f32x4* wg = (const f32x4*)wpack;
    const int lane = lane_id(), wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;

    lds_fill(bias_l, bias, TL_NBIAS);
    stream_prologue<CH4>(wg, buf0);
    stream_wait();
    __syncthreads();
    f32x4 *cur = buf0, *nxt = buf1;

    const int ntiles = (int)((Q + NW * 16 - 1) / (NW * 16));
    int first, count, stride;
    xcd_tile_range(ntiles, first, count, stride);
    for (int it = 0; it < count; ++it) {
        const int64_t qi = (int64_t)(first + it * stride) * (NW * 16) + wave * 16 + n;
        const bool qv = qi < Q;
        const int64_t qc = qv ? qi : Q - 1;
        f32x4 p[16], x[8], h[16];
        {
            const f32x4* sp = (const f32x4*)(pooled + qc * 256) + g;
            const f32x4* sx = (const f32x4*)(xbar + qc * 128) + g;
#pragma unroll
            for (int bb = 0; bb < 16; ++bb) p[bb] = sp[4 * bb];
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) x[bb] = sx[4 * bb];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)                                  // Wa . pooled + bias: chunks 0..7 (two output blocks each)
            stream_step<CH4>(wg + (c + 1) * CH4, cur, nxt,
                           [&](const f32x4* w) { dense_blocks<16, 2, 0>(p, &h[2 * c], w, bias4 + 8 * c, lane); });
#pragma unroll
        for (int c = 0; c < 4; ++c)                                  // + Wb . xbar (K = 128: four output blocks per 32 KiB chunk), ReLU: chunks 8..11
            stream_step<CH4>(wg + (c + 9) * CH4, cur, nxt,
                           [&](const f32x4* w) { dense_blocks<8, 4, 1, 1>(x, &h[4 * c], w, bias4, lane); });
#pragma unroll
        for (int c = 0; c < 8; ++c)                                  // L2: chunks 12..19, then L3 (chunk 20)
            stream_step<CH4>(wg + (c + 13) * CH4, cur, nxt,
                           [&](const f32x4* w) { dense_blocks<16, 2, 1>(h, &p[2 * c], w, bias4 + 64 + 8 * c, lane); });
        f32x4 o[2];
        stream_step<CH4>(wg, cur, nxt, [&](const f32x4* w) { dense_blocks<16, 2, 0>(p, o, w, bias4 + 128, lane); });
        if (qv && g == 0) {
            const float l0 = o[0].x, l1 = o[0].y;
            logits[qi * 2] = l0;
            logits[qi * 2 + 1] = l1;
            if (occ) {
                const float mx = fmaxf(l0, l1);
                const float e0 = __expf(l0 - mx), e1 = __expf(l1 - mx);
                occ[qi] = (e0 - e1) / (e0 + e1);
            }
        }
    }
}

// The tail in split precision ("f16x3"): the three dense layers as f16 hi/lo products (dense_blocks_f16x3); the 512 -> 256 layer is the sum of
// two 256 -> 256 halves ([pooled | xbar] are two tensors), the second starts from the fp32 accumulators of the first.
// w16 (half8 fragments): pps_pack_dense_f16x3 images, 32 KiB chunks (two output blocks each): Wa and Wb ALTERNATING chunk by chunk (the pair of
// output blocks 2c, 2c+1 is finished before the next one starts: two live accumulator blocks instead of sixteen), then [L2][L3].
__global__ __launch_bounds__(NT, 2) void decode_tail_h_kernel(const float* __restrict__ pooled, const float* __restrict__ xbar, int64_t Q,
                                                              const f32x4* __restrict__ w16, const float* __restrict__ bias,
                                                              float* __restrict__ logits, float* __restrict__ occ, int* __restrict__ range) {
    float amax = 0.f;
    f32x4* buf0 = (f32x4*)pps_smem;
    f32x4* buf1 = buf0 + CH4;
    float* bias_l = (float*)(buf1 + CH4);
    const f32x4* bias4 = (const f32x4*)bias_l;
    const f32x4* wg = w16;
    const int lane = lane_id(), wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;

    lds_fill(bias_l, bias, TL_NBIAS);
    stream_prologue<CH4>(wg, buf0);
    stream_wait();
    __syncthreads();
    f32x4 *cur = buf0, *nxt = buf1;

    const int ntiles = (int)((Q + NW * 16 - 1) / (NW * 16));
    int first, count, stride;
    xcd_tile_range(ntiles, first, count, stride);
    for (int it = 0; it < count; ++it) {
        const int64_t qi = (int64_t)(first + it * stride) * (NW * 16) + wave * 16 + n;
        const bool qv = qi < Q;
        const int64_t qc = qv ? qi : Q - 1;
        HiLo p[8], x[4];
        {
            const f32x4* sp = (const f32x4*)(pooled + qc * 256) + g;
            const f32x4* sx = (const f32x4*)(xbar + qc * 128) + g;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) p[kb] = split_f16_r(amax, sp[4 * (2 * kb)], sp[4 * (2 * kb + 1)]);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) x[kb] = split_f16_r(amax, sx[4 * (2 * kb)], sx[4 * (2 * kb + 1)]);
        }
        HiLo y[8];
        // image: 8 x [Wa pair c: 2048 f32x4 | Wb pair c: 1024 f32x4], L2 at 24576 (8 x 2048), L3 at 40960
#pragma unroll
        for (int c = 0; c < 8; ++c) {                                  // output blocks 2c, 2c+1 of the (256 + 128) -> 256 layer: the chunks of Wa and Wb alternate
            f32x4 h[2];
            stream_step<CH4 / 2>(wg + c * 3072 + 2048, cur, nxt, [&](const f32x4* w) {           // Wa . pooled + bias, no activation yet; next: Wb pair c (16 KiB)
                dense_blocks_f16x3<8, 2, 0>(p, (const half8*)w, bias4 + 8 * c, lane, [&](int, const f32x4& o0, const f32x4& o1) { h[0] = o0; h[1] = o1; }); });
            stream_step<CH4>(c + 1 < 8 ? wg + (c + 1) * 3072 : wg + 24576, cur, nxt, [&](const f32x4* w) {      // + Wb . xbar (K = 128), ReLU
                dense_blocks_f16x3<4, 2, 1, true, true>(x, (const half8*)w, bias4, lane, [&](int, const f32x4& o0, const f32x4& o1) { y[c] = split_f16_r(amax, o0, o1); }, h); });
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)
            stream_step<CH4>(wg + 24576 + (c + 1) * 2048, cur, nxt, [&](const f32x4* w) {
                dense_blocks_f16x3<8, 2, 1>(y, (const half8*)w, bias4 + 64 + 8 * c, lane, [&](int, const f32x4& o0, const f32x4& o1) { p[c] = split_f16_r(amax, o0, o1); }); });
        f32x4 o[2];
        stream_step<CH4>(wg, cur, nxt, [&](const f32x4* w) {
            dense_blocks_f16x3<8, 2, 0>(p, (const half8*)w, bias4 + 128, lane, [&](int, const f32x4& o0, const f32x4& o1) { o[0] = o0; o[1] = o1; }); });
        if (qv && g == 0) {
            const float l0 = o[0].x, l1 = o[0].y;
            logits[qi * 2] = l0;
            logits[qi * 2 + 1] = l1;
            if (occ) {
                const float mx = fmaxf(l0, l1);
                const float e0 = __expf(l0 - mx), e1 = __expf(l1 - mx);
                occ[qi] = (e0 - e1) / (e0 + e1);
            }
        }
    }
    range_commit(amax, range);
}

// =====================================================================================================
// host side
// =====================================================================================================
static int g_cu_count = 0;
static int cu_count() {
    if (g_cu_count == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        g_cu_count = prop.multiProcessorCount;
    }
    return g_cu_count;
}

template <class K>
static int set_lds(K kernel, int bytes) {
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 0 : 1;
}

static int grid_for(int64_t ntiles) {
    int cus = cu_count();
    if (cus <= 0) cus = 256;
    cus *= WG_PER_CU;
    return (int)(ntiles < cus ? (ntiles > 0 ? ntiles : 1) : cus);
}

// PointNet row kernels: PN_WG_PER_CU workgroups of PNW waves per CU.  Packed wave groups (qg queries, no padded rows) are used for as many
// FULL rounds over all waves of the chip as the query count allows; the remainder runs one query per wave (patch_packing mode 2: the packed
// arithmetic, bit-identical per query) so that the last round is short instead of a whole packed group (Q = 50000, P = 50: 3 x 25 + 4 tiles
// per wave instead of 100).
struct PnSplit { int64_t q_packed; int grid_packed, grid_rest, rest_mode; };
static PnSplit pn_split(int64_t q, int p) {
    int cus = cu_count();
    if (cus <= 0) cus = 256;
    const PatchPacking pk = patch_packing(p);
    PnSplit sp;
    const int wgs = cus * PN_WG_PER_CU;
    const int64_t per_round = (int64_t)wgs * PNW * pk.qg;
    sp.q_packed = pk.packed ? (q / per_round) * per_round : 0;
    if (pk.packed && getenv("PPS_PN_FORCE_PACK")) sp.q_packed = q;      // test hook: packed path for any query count
    sp.grid_packed = wgs;
    sp.rest_mode = pk.packed ? 2 : 0;                // the remainder in the packed arithmetic, one query per wave group: same bits as a packed round
    const int64_t rest_tiles = (q - sp.q_packed + PNW - 1) / PNW;
    sp.grid_rest = (int)(rest_tiles < wgs ? rest_tiles : wgs);
    return sp;
}

#define PPS_LAUNCH_CHECK() (hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH)

template <bool H>
static int launch_stn_rows(const float* patches, int64_t q, int p, const float* wpack, const void* wdense, const float* bias, float* g,
                           int* flag, void* stream) {
    static int once = set_lds(pointnet_stn_rows_kernel<H>, PA_LDS_BYTES);
    (void)once;
    const PnSplit sp = pn_split(q, p);
    if (sp.q_packed > 0)
        hipLaunchKernelGGL(pointnet_stn_rows_kernel<H>, dim3(sp.grid_packed), dim3(PNT), PA_LDS_BYTES, (hipStream_t)stream, patches,
                           sp.q_packed, p, 1, wpack, (const f32x4*)wdense, bias, g, flag);
    if (q > sp.q_packed)
        hipLaunchKernelGGL(pointnet_stn_rows_kernel<H>, dim3(sp.grid_rest), dim3(PNT), PA_LDS_BYTES, (hipStream_t)stream,
                           patches + sp.q_packed * p * 3, q - sp.q_packed, p, sp.rest_mode, wpack, (const f32x4*)wdense, bias, g + sp.q_packed * 256, flag);
    return PPS_LAUNCH_CHECK();
}

template <bool H>
static int launch_feat_rows(const float* patches, const float* trans2, int64_t q, int p, const float* wpack, const void* wdense,
                            const float* bias, float* xbar, int* flag, void* stream) {
    static int once = set_lds(pointnet_feat_rows_kernel<H>, PC_LDS_BYTES);
    (void)once;
    int cus = cu_count();
    if (cus <= 0) cus = 256;
    const int64_t ntiles = (q + PNW - 1) / PNW, wgs = (int64_t)cus * PN_WG_PER_CU;
    hipLaunchKernelGGL(pointnet_feat_rows_kernel<H>, dim3((unsigned)(ntiles < wgs ? ntiles : wgs)), dim3(PNT), PC_LDS_BYTES, (hipStream_t)stream, patches,
                       trans2, q, p, wpack, (const f32x4*)wdense, bias, xbar, flag);
    return PPS_LAUNCH_CHECK();
}


// ---- launchers with the range-guard plumbing (the extern "C" entry points below are thin wrappers) ----------------------------------------------
// gate (fp32 kernels): null = run; otherwise the launch returns at once unless gate[0] != 0.  range (split-precision kernels): where a kernel reports
// that an activation left the f16 range (null = not reported).
static int interp_pool_f32_impl(const float* G, const float* pts, const float* query, const int64_t* idx, int64_t q, int k,
                                const float* wpack, const float* bias, float* pooled, const int* gate, void* stream) {
    if (q < 0 || k < 1 || k > 64) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!G || !pts || !query || !idx || !wpack || !bias || !pooled) return PPS_ERR_ARG;
    static int once = set_lds(interp_pool_kernel, IP_LDS_BYTES);
    (void)once;
    hipLaunchKernelGGL(interp_pool_kernel, dim3(grid_for((q + NW / 4 - 1) / (NW / 4))), dim3(NT), IP_LDS_BYTES, (hipStream_t)stream,
                       G, pts, query, idx, q, k, wpack, bias, pooled, gate);
    return PPS_LAUNCH_CHECK();
}

static int stn_fc_f32_impl(const float* g, int64_t q, const float* wpack, const float* bias, float* trans2, const int* gate, void* stream) {
    if (q < 0) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!g || !wpack || !bias || !trans2) return PPS_ERR_ARG;
    static int once = set_lds(pointnet_stn_fc_kernel, PB_LDS_BYTES);
    (void)once;
    hipLaunchKernelGGL(pointnet_stn_fc_kernel, dim3(grid_for((q + NW * 16 * PB_T - 1) / (NW * 16 * PB_T))), dim3(NT), PB_LDS_BYTES, (hipStream_t)stream,
                       g, q, wpack, bias, trans2, gate);
    return PPS_LAUNCH_CHECK();
}

static int decode_tail_f32_impl(const float* pooled, const float* xbar, int64_t q, const float* wpack, const float* bias,
                                float* logits, float* occ, int* gate, void* stream) {
    if (q < 0) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!pooled || !xbar || !wpack || !bias || !logits) return PPS_ERR_ARG;
    static int once = set_lds(decode_tail_kernel, TL_LDS_BYTES);
    (void)once;
    hipLaunchKernelGGL(decode_tail_kernel, dim3(grid_for((q + NW * 16 - 1) / (NW * 16))), dim3(NT), TL_LDS_BYTES, (hipStream_t)stream,
                       pooled, xbar, q, wpack, bias, logits, occ, gate);
    return PPS_LAUNCH_CHECK();
}

extern "C" {

int pps_abi_version(void) { return 2; }

// development aid (not part of the public header): resident workgroups per CU of the decoder kernels
int pps_debug_occupancy(int which) {
    int n = -1;
    if (which == 0) { set_lds(interp_pool_kernel, IP_LDS_BYTES); hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, interp_pool_kernel, NT, IP_LDS_BYTES); }
    if (which == 1) { set_lds(pointnet_stn_rows_kernel<false>, PA_LDS_BYTES); hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pointnet_stn_rows_kernel<false>, PNT, PA_LDS_BYTES); }
    if (which == 3) { set_lds(pointnet_stn_fc_kernel, PB_LDS_BYTES); hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pointnet_stn_fc_kernel, NT, PB_LDS_BYTES); }
    if (which == 4) { set_lds(pointnet_stn_fc_h_kernel, PB_LDS_BYTES); hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pointnet_stn_fc_h_kernel, NT, PB_LDS_BYTES); }
    if (which == 2) { set_lds(pointnet_feat_rows_kernel<false>, PC_LDS_BYTES); hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pointnet_feat_rows_kernel<false>, PNT, PC_LDS_BYTES); }
    return n;
}
int pps_device_cu_count(void) { return cu_count(); }

int pps_rows_dense256_f32(const float* in, int64_t rs, int64_t cs, int64_t m, const float* wpack, const float* bias,
                          float* out, void* stream) {
    if (m < 0) return PPS_ERR_ARG;
    if (m == 0) return PPS_OK;
    if (!in || !wpack || !bias || !out) return PPS_ERR_ARG;
    if (cs == 1 && (rs % 4) != 0) return PPS_ERR_ARG;
    static int once = set_lds(rows_dense256_kernel, RD_LDS_BYTES);
    (void)once;
    hipLaunchKernelGGL(rows_dense256_kernel, dim3(grid_for((m + NW * 16 - 1) / (NW * 16))), dim3(NT), RD_LDS_BYTES, (hipStream_t)stream,
                       in, rs, cs, m, wpack, bias, out);
    return PPS_LAUNCH_CHECK();
}

int pps_interp_pool_f32(const float* G, const float* pts, const float* query, const int64_t* idx, int64_t q, int k,
                        const float* wpack, const float* bias, float* pooled, void* stream) {
    return interp_pool_f32_impl(G, pts, query, idx, q, k, wpack, bias, pooled, nullptr, stream);
}

int pps_interp_pool_f16x3(const float* G, const float* pts, const float* query, const int64_t* idx, int64_t q, int k,
                          const float* wxyz, const void* w16, const float* bias, float* pooled, int32_t* range_flag, void* stream) {
    if (q < 0 || k < 1 || k > 64) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!G || !pts || !query || !idx || !wxyz || !w16 || !bias || !pooled) return PPS_ERR_ARG;
    static int once = set_lds(interp_pool_f16x3_kernel, IH_LDS_BYTES);
    (void)once;
    int cus = cu_count();
    if (cus <= 0) cus = 256;
    const int64_t ntiles = (q + IH_NW / 4 - 1) / (IH_NW / 4);
    const int64_t wgs = (int64_t)cus * IH_WG_PER_CU;                  // one 8-wave workgroup per CU (IH_NT = 512)
    const int grid = (int)(ntiles < wgs ? ntiles : wgs);
    hipLaunchKernelGGL(interp_pool_f16x3_kernel, dim3(grid), dim3(IH_NT), IH_LDS_BYTES, (hipStream_t)stream,
                       G, pts, query, idx, q, k, wxyz, (const f32x4*)w16, bias, pooled, (int*)range_flag);
    return PPS_LAUNCH_CHECK();
}

int pps_interp_small_f32(const float* G, const float* pts, const float* query, const int64_t* idx, int64_t q, int k, int c,
                         const float* wpack, const float* bias, const float* wtail, int nout, float* out, void* stream) {
    if (q < 0 || k < 1 || k > 64 || (c != 32 && c != 64) || nout < 1 || nout > IS_MAX_OUT) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!G || !pts || !query || !idx || !wpack || !bias || !wtail || !out) return PPS_ERR_ARG;
    int cus = cu_count();
    if (cus <= 0) cus = 256;
    const int grid = (int)(q < (int64_t)cus * 8 ? q : (int64_t)cus * 8);
    hipStream_t st = (hipStream_t)stream;
    if (c == 32) hipLaunchKernelGGL((interp_small_kernel<2, false>), dim3(grid), dim3(IS_NT), 0, st, G, pts, query, idx, q, k, wpack, nullptr, bias, wtail, nout, out, nullptr);
    else hipLaunchKernelGGL((interp_small_kernel<4, false>), dim3(grid), dim3(IS_NT), 0, st, G, pts, query, idx, q, k, wpack, nullptr, bias, wtail, nout, out, nullptr);
    return PPS_LAUNCH_CHECK();
}

int pps_interp_small_f16x3(const float* G, const float* pts, const float* query, const int64_t* idx, int64_t q, int k, int c,
                           const float* wpack, const void* w16, const float* bias, const float* wtail, int nout, float* out, int* guard,
                           void* stream) {
    if (q < 0 || k < 1 || k > 64 || (c != 32 && c != 64) || nout < 1 || nout > IS_MAX_OUT) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!G || !pts || !query || !idx || !wpack || !w16 || !bias || !wtail || !out || !guard) return PPS_ERR_ARG;
    int cus = cu_count();
    if (cus <= 0) cus = 256;
    const int grid = (int)(q < (int64_t)cus * 8 ? q : (int64_t)cus * 8);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(guard, 0, sizeof(int), st) != hipSuccess) return PPS_ERR_LAUNCH;
    const float* wh = (const float*)w16;
    // split precision first; the fp32 kernel behind it runs only if an activation left the f16 range (no host round trip)
    if (c == 32) {
        hipLaunchKernelGGL((interp_small_kernel<2, true>), dim3(grid), dim3(IS_NT), 0, st, G, pts, query, idx, q, k, wpack, wh, bias, wtail, nout, out, guard);
        hipLaunchKernelGGL((interp_small_kernel<2, false>), dim3(grid), dim3(IS_NT), 0, st, G, pts, query, idx, q, k, wpack, nullptr, bias, wtail, nout, out, guard);
    } else {
        hipLaunchKernelGGL((interp_small_kernel<4, true>), dim3(grid), dim3(IS_NT), 0, st, G, pts, query, idx, q, k, wpack, wh, bias, wtail, nout, out, guard);
        hipLaunchKernelGGL((interp_small_kernel<4, false>), dim3(grid), dim3(IS_NT), 0, st, G, pts, query, idx, q, k, wpack, nullptr, bias, wtail, nout, out, guard);
    }
    return PPS_LAUNCH_CHECK();
}

int pps_pointnet_stn_rows_f32(const float* patches, int64_t q, int p, const float* wpack, const float* bias, float* g,
                              void* stream) {
    if (q < 0 || p < 1) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!patches || !wpack || !bias || !g) return PPS_ERR_ARG;
    return launch_stn_rows<false>(patches, q, p, wpack, wpack + PA_W_XYZ, bias, g, nullptr, stream);
}

int pps_pointnet_stn_fc_f32(const float* g, int64_t q, const float* wpack, const float* bias, float* trans2, void* stream) {
    return stn_fc_f32_impl(g, q, wpack, bias, trans2, nullptr, stream);
}

int pps_pointnet_feat_rows_f32(const float* patches, const float* trans2, int64_t q, int p, const float* wpack,
                               const float* bias, float* xbar, void* stream) {
    if (q < 0 || p < 1) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!patches || !trans2 || !wpack || !bias || !xbar) return PPS_ERR_ARG;
    return launch_feat_rows<false>(patches, trans2, q, p, wpack, wpack + PC_W_XYZ, bias, xbar, nullptr, stream);
}

/* Split-precision PointNet branch ("f16x3"): w16[0..2] = f16x3 images of (c0b, s1, s2, s3), (fc1, fc2, fc3), (c0b, c1, c2, c3); the fp32
 * images supply the xyz layers and the biases.  trans2 (q x 16 KiB of scratch) holds the pre-split fragments between the kernels. */
int pps_pointnet_f16x3(const float* patches, int64_t q, int p, const float* const* weights /* [2..7] of the decode array */,
                       const void* const* w16, float* g, float* trans2, float* xbar, int32_t* range_flag, void* const* events, void* stream) {
    if (q < 0 || p < 1) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!patches || !weights || !w16 || !w16[0] || !w16[1] || !w16[2] || !g || !trans2 || !xbar) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int* range = (int*)range_flag;
    int rc = launch_stn_rows<true>(patches, q, p, weights[0], w16[0], weights[1], g, range, stream);
    if (events && events[0]) hipEventRecord((hipEvent_t)events[0], st);
    if (rc != PPS_OK) return rc;
    static int once = set_lds(pointnet_stn_fc_h_kernel, PB_LDS_BYTES);
    (void)once;
    // The per-query matrix M (16 KiB per query) crosses memory between these two kernels.  `sub` > 0: the pair runs sub-chunk by sub-chunk
    // (M of a sub-chunk = sub x 16 KiB, written and read back to back; with `reuse` every sub-chunk uses the SAME scratch addresses) --
    // the measured answer to "does a cache-resident M help" (tools/time_pn_subchunks.py).  Results are bit-identical:
    // both kernels treat queries independently.
    static const int64_t sub = getenv("PPS_PN_SUB") ? atoll(getenv("PPS_PN_SUB")) : 0;
    static const bool reuse = getenv("PPS_PN_SUB_REUSE") != nullptr;
    if (sub > 0 && sub < q) {
        for (int64_t off = 0; off < q && rc == PPS_OK; off += sub) {
            const int64_t n = q - off < sub ? q - off : sub;
            float* t2 = reuse ? trans2 : trans2 + off * 4096;
            hipLaunchKernelGGL(pointnet_stn_fc_h_kernel, dim3(grid_for((n + NW * 16 - 1) / (NW * 16))), dim3(NT), PB_LDS_BYTES, st, g + off * 256, n,
                               (const f32x4*)w16[1], weights[3], (half8*)t2, range);
            rc = launch_feat_rows<true>(patches + off * p * 3, t2, n, p, weights[4], w16[2], weights[5], xbar + off * 128, range, stream);
        }
        if (events && events[1]) hipEventRecord((hipEvent_t)events[1], st);
        if (events && events[2]) hipEventRecord((hipEvent_t)events[2], st);
        return rc;
    }
    hipLaunchKernelGGL(pointnet_stn_fc_h_kernel, dim3(grid_for((q + NW * 16 - 1) / (NW * 16))), dim3(NT), PB_LDS_BYTES, st, g, q,
                       (const f32x4*)w16[1], weights[3], (half8*)trans2, range);
    if (events && events[1]) hipEventRecord((hipEvent_t)events[1], st);
    if (hipGetLastError() != hipSuccess) return PPS_ERR_LAUNCH;
    rc = launch_feat_rows<true>(patches, trans2, q, p, weights[4], w16[2], weights[5], xbar, range, stream);
    if (events && events[2]) hipEventRecord((hipEvent_t)events[2], st);
    return rc;
}

int pps_decode_tail_f32(const float* pooled, const float* xbar, int64_t q, const float* wpack, const float* bias,
                        float* logits, float* occ, void* stream) {
    return decode_tail_f32_impl(pooled, xbar, q, wpack, bias, logits, occ, nullptr, stream);
}

int pps_decode_tail_f16x3(const float* pooled, const float* xbar, int64_t q, const void* w16, const float* bias, float* logits, float* occ,
                          int32_t* range_flag, void* stream) {
    if (q < 0) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!pooled || !xbar || !w16 || !bias || !logits) return PPS_ERR_ARG;
    static int once = set_lds(decode_tail_h_kernel, TL_LDS_BYTES);
    (void)once;
    hipLaunchKernelGGL(decode_tail_h_kernel, dim3(grid_for((q + NW * 16 - 1) / (NW * 16))), dim3(NT), TL_LDS_BYTES, (hipStream_t)stream,
                       pooled, xbar, q, (const f32x4*)w16, bias, logits, occ, (int*)range_flag);
    return PPS_LAUNCH_CHECK();
}

/* The whole decoder of one query chunk in one call: the five launches above on `stream`, intermediates in caller scratch (the first 64 bytes of it
 * hold the range-guard words of the split-precision path: [0] flag of the current chunk, [1] number of chunks that fell back to fp32). */
#define PPS_DECODE_WS_HEAD 64
size_t pps_decode_ws_bytes(int64_t q) { return q < 0 ? 0 : (size_t)q * (256 + 256 + 4096 + 256) * sizeof(float) + PPS_DECODE_WS_HEAD; }

static int decode_fwd(const float* table, const float* pts, const float* query, const int64_t* idx, int64_t q, int k,
                      const float* patches, int p, const float* const* weights, float* logits, float* occ, void* ws, void* const* events,
                      void* stream, const void* const* w16 = nullptr) {
    const void* interp_w16 = w16 ? w16[0] : nullptr;
    const bool pn16 = w16 && w16[1] && w16[2] && w16[3];
    if (q < 0) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!weights || !ws || !logits) return PPS_ERR_ARG;
    for (int i = 0; i < 10; ++i)
        if (!weights[i]) return PPS_ERR_ARG;
    int* guard = (int*)ws;
    float* pooled = (float*)ws + PPS_DECODE_WS_HEAD / sizeof(float);
    float* g = pooled + q * 256;
    float* trans2 = g + q * 256;
    float* xbar = trans2 + q * 4096;
    hipStream_t st = (hipStream_t)stream;
    const bool split = interp_w16 || pn16 || (w16 && w16[4]);
    // split precision: the chunk's range flag starts at zero (the fall-back counter guard[1] is left alone)
    if (split && hipMemsetAsync(guard, 0, sizeof(int), st) != hipSuccess) return PPS_ERR_LAUNCH;
#define PPS_MARK(i) do { if (events && events[i] && hipEventRecord((hipEvent_t)events[i], st) != hipSuccess) return PPS_ERR_LAUNCH; } while (0)
    PPS_MARK(0);
    int rc = interp_w16 ? pps_interp_pool_f16x3(table, pts, query, idx, q, k, weights[0], interp_w16, weights[1], pooled, guard, stream)
                        : pps_interp_pool_f32(table, pts, query, idx, q, k, weights[0], weights[1], pooled, stream);
    PPS_MARK(1);
    if (pn16) {
        if (rc == PPS_OK) rc = pps_pointnet_f16x3(patches, q, p, weights + 2, w16 + 1, g, trans2, xbar, guard, events ? events + 2 : nullptr, stream);
    } else {
        if (rc == PPS_OK) rc = pps_pointnet_stn_rows_f32(patches, q, p, weights[2], weights[3], g, stream);
        PPS_MARK(2);
        if (rc == PPS_OK) rc = pps_pointnet_stn_fc_f32(g, q, weights[4], weights[5], trans2, stream);
        PPS_MARK(3);
        if (rc == PPS_OK) rc = pps_pointnet_feat_rows_f32(patches, trans2, q, p, weights[6], weights[7], xbar, stream);
        PPS_MARK(4);
    }
    if (rc == PPS_OK)
        rc = (w16 && w16[4]) ? pps_decode_tail_f16x3(pooled, xbar, q, w16[4], weights[9], logits, occ, guard, stream)
                             : pps_decode_tail_f32(pooled, xbar, q, weights[8], weights[9], logits, occ, stream);
    PPS_MARK(5);
#undef PPS_MARK
    if (split && rc == PPS_OK) {
        // Range guard: the exact-fp32 kernels of the same chunk, each returning at once unless a split-precision kernel above raised guard[0]
        // (an activation beyond +-65504, which f16(x) cannot hold).  Normal chunks pay five empty launches (~12 us of 5 ms); a chunk that left
        // the range is recomputed in fp32 on the device, no host round trip, and counted in guard[1].
        rc = interp_pool_f32_impl(table, pts, query, idx, q, k, weights[0], weights[1], pooled, guard, stream);
        if (rc == PPS_OK) rc = launch_stn_rows<false>(patches, q, p, weights[2], weights[2] + PA_W_XYZ, weights[3], g, guard, stream);
        if (rc == PPS_OK) rc = stn_fc_f32_impl(g, q, weights[4], weights[5], trans2, guard, stream);
        if (rc == PPS_OK) rc = launch_feat_rows<false>(patches, trans2, q, p, weights[6], weights[6] + PC_W_XYZ, weights[7], xbar, guard, stream);
        if (rc == PPS_OK) rc = decode_tail_f32_impl(pooled, xbar, q, weights[8], weights[9], logits, occ, guard, stream);
    }
    return rc;
}

int pps_decode_fwd_f32(const float* table, const float* pts, const float* query, const int64_t* idx, int64_t q, int k,
                       const float* patches, int p, const float* const* weights, float* logits, float* occ, void* ws, void* stream) {
    return decode_fwd(table, pts, query, idx, q, k, patches, p, weights, logits, occ, ws, nullptr, stream);
}

int pps_decode_fwd_events_f32(const float* table, const float* pts, const float* query, const int64_t* idx, int64_t q, int k,
                              const float* patches, int p, const float* const* weights, float* logits, float* occ, void* ws,
                              void* const* events, void* stream) {
    return decode_fwd(table, pts, query, idx, q, k, patches, p, weights, logits, occ, ws, events, stream);
}

int pps_decode_fwd_mixed_f32(const float* table, const float* pts, const float* query, const int64_t* idx, int64_t q, int k,
                             const float* patches, int p, const float* const* weights, const void* const* w16, float* logits, float* occ,
                             void* ws, void* const* events, void* stream) {
    if (!w16) return PPS_ERR_ARG;
    return decode_fwd(table, pts, query, idx, q, k, patches, p, weights, logits, occ, ws, events, stream, w16);
}

}  // extern "C"
