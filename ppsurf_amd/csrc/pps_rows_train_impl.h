// Implementation of pps_rows_train.hip for ONE 16-bit storage type; included once per type inside its own namespace (PPS_NS), with
// PPS_ELEM_F16 = 0 (bfloat16) or 1 (IEEE half).  The functions named like the C entries are the per-type bodies the real entries dispatch to.
namespace PPS_NS {


typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// 16-bit storage element of this instantiation (PPS_ELEM_F16: IEEE half for trainer.precision 16-mixed; else bfloat16)
#if PPS_ELEM_F16
typedef _Float16 elem8 __attribute__((ext_vector_type(8)));
typedef _Float16 elem2 __attribute__((ext_vector_type(2)));
#define PPS_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_f16
__device__ __forceinline__ float lo16(unsigned u) { return (float)(*(const elem2*)&u)[0]; }
__device__ __forceinline__ float hi16(unsigned u) { return (float)(*(const elem2*)&u)[1]; }
#else
typedef __bf16 elem8 __attribute__((ext_vector_type(8)));
typedef __bf16 elem2 __attribute__((ext_vector_type(2)));
#define PPS_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
__device__ __forceinline__ float lo16(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi16(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
#endif
typedef elem8 bf16x8;                   // (historic name in the kernels below: 8 storage elements = one MFMA operand)

constexpr int NT = 512;                 // 8 waves
constexpr int NWAVE = NT / 64;
constexpr int MAXP = 256;               // most workgroups (= slab partials) of a launch

__device__ __forceinline__ unsigned pack2(float a, float b) {                     // round to nearest even (v_cvt_pk_bf16_f32 / v_cvt_f16_f32)
    const f32x2 v = {a, b};
    const elem2 h = __builtin_convertvector(v, elem2);
    return *(const unsigned*)&h;
}
__device__ __forceinline__ bf16x8 as_frag(const u32x4& u) { return *(const bf16x8*)&u; }

// ReLU of 8 packed elements: negative values (sign bit set) are negative as int16 too (both formats are sign-magnitude): v_pk_max_i16 with 0
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu_bf16x2(unsigned u) {
    const s16x2 v = __builtin_elementwise_max(*(const s16x2*)&u, s16x2{0, 0});
    return *(const unsigned*)&v;
}
__device__ __forceinline__ u32x4 relu_bf16x8(const u32x4& q) { return u32x4{relu_bf16x2(q.x), relu_bf16x2(q.y), relu_bf16x2(q.z), relu_bf16x2(q.w)}; }

// output channel of MFMA row m = 4 g + r of 16-row block ob: lane (row n, g) then holds, over the 4 blocks of a group and r = 0..3,
// the 16 consecutive channels 64 (ob >> 2) + 16 g + [4 (ob & 3) + r]
__device__ __forceinline__ void block_row_of(int c, int& ob, int& m) {
    const int rem = c & 63;
    ob = 4 * (c >> 6) + ((rem >> 2) & 3);
    m = 4 * (rem >> 4) + (rem & 3);
}

__device__ __forceinline__ float row16_sum(float v) {                            // over the 16 lanes that share lane >> 4
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) v += __shfl_xor(v, s);
    return v;
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// ---------------------------------------------------------------------------------------------------------------------
// forward / input gradient: one kernel body.
//   CK   channels contracted (fwd: cin, dx: cout)       CO  channels produced (fwd: cout, dx: cin)
//   DX   false: forward -- B operand = act(x), epilogue = + bias, store y, statistics of y
//        true : dx      -- B operand = G = gy + gS + gQ2 * y, epilogue = mask, * scale_in, store dx, sums for d scale_in / d shift_in
// ---------------------------------------------------------------------------------------------------------------------
struct LayerArgs {
    const uint16_t* src;          // fwd: x [rows, CK]            dx: gy [rows, CK]
    const uint16_t* src2;         // fwd: -                       dx: y [rows, CK] or NULL (no statistics on the output)
    const float* pre_a;           // fwd: scale_in [CK] or NULL   dx: gS [CK] or NULL
    const float* pre_b;           // fwd: shift_in [CK] or NULL   dx: 2 gQ [CK] or NULL
    int pre_relu;                 // fwd: relu on the input
    const float* w;               // [cout, cin] fp32 master weights
    const float* bias;            // fwd: [CO] or NULL
    const uint16_t* xin;          // dx: x [rows, CO] (mask, d scale_in) or NULL (identity input)
    const float* in_scale;        // dx: [CO] or NULL
    const float* in_shift;        // dx: [CO] or NULL
    int in_relu;
    const uint16_t* addend;       // dx: [rows, CO] added to the result (the other gradient of x; may be dst itself) or NULL
    uint16_t* dst;                // fwd: y [rows, CO]            dx: dx [rows, CO]
    float* partials;              // [gridDim.x][2][CO] or NULL: fwd (sum y, sum y^2); dx (sum d * x, sum d)
    int64_t rows;
    const uint8_t* garg;          // dx, POOL: winning row of every (group, channel) [rows / pool_p, CK]; src is then the gradient per GROUP
    int pool_p;                   // dx, POOL: rows per group
    unsigned pool_magic;          // floor(2^32 / pool_p) + 1
    // dx, POOL == 2: src is not read -- the gradient of a layer whose raw output only feeds the single-head attention pooling over the pool_p rows
    // of every group (PointNet's AttentionPoco, pps_patch_attn_bwd_weights) has rank two per group and is rebuilt on load (att_grad8)
    const float* g_a;             // [rows]  softmax weight of the row
    const float* g_dl;            // [rows]  gradient of the row's logit
    const float* g_dp;            // [rows / pool_p, CK] fp32: gradient of the pooled row
    const float* g_v;             // [CK] fp32: the logit's weight vector
    // dx, second gradient of x rebuilt instead of read (attention pooling over the att_k rows of a group, pps_attn_pool_bwd_weights):
    //   dx[row, c] += relu'(x[row, c]) * att_a[row] * att_dp[row / att_k, c]
    const float* att_a;           // [rows] or NULL
    const uint16_t* att_dp;       // [rows / att_k, CO]
    int att_k;
    unsigned att_magic;           // floor(2^32 / att_k) + 1 (att_k >= 2; one row per group needs no division)
};

// The incoming gradient of a layer whose output only feeds a max over the p rows of every group (the STN of PointNet, source/base/nn.py:181) is
// one value per (group, channel), placed in the winning row: 8 channels of row `row` are rebuilt from the group's 16 bytes of gradient and
// 8 bytes of winners (both stay in the caches: every group is read by its p rows) instead of streaming a [rows, C] tensor that is 98 % zeros.
// row / pool_p as a multiply-high with magic = floor(2^32 / p) + 1: exact while rows * p < 2^32 (checked by the host), no division sequence
// in the load path of a kernel whose staging arithmetic is what limits it
__device__ __forceinline__ u32x4 pooled_grad8(const uint16_t* gval, const uint8_t* garg, int pool_p, unsigned magic, int64_t row, int c, int col8) {
    const unsigned grp = __umulhi((unsigned)row, magic), r = (unsigned)row - grp * (unsigned)pool_p;
    const u32x4 d = *(const u32x4*)(gval + (int64_t)grp * c + col8);
    const u32x2 w = *(const u32x2*)(garg + (int64_t)grp * c + col8);
    u32x4 o;
    o.x = ((w.x & 0xffu) == r ? d.x & 0xffffu : 0u) | (((w.x >> 8) & 0xffu) == r ? d.x & 0xffff0000u : 0u);
    o.y = (((w.x >> 16) & 0xffu) == r ? d.y & 0xffffu : 0u) | ((w.x >> 24) == r ? d.y & 0xffff0000u : 0u);
    o.z = ((w.y & 0xffu) == r ? d.z & 0xffffu : 0u) | (((w.y >> 8) & 0xffu) == r ? d.z & 0xffff0000u : 0u);
    o.w = (((w.y >> 16) & 0xffu) == r ? d.w & 0xffffu : 0u) | ((w.y >> 24) == r ? d.w & 0xffff0000u : 0u);
    return o;
}

// gy[row, c] = a[row] * dP[row / p, c] + dl[row] * v[c]  (pps_attn_train.hip patch_attn_bwd_kernel: the same two products and one sum in fp32, rounded
// to the storage type like the tensor it used to store): 8 channels of one row from 2 floats of the row, 32 cached bytes of its group and of v
__device__ __forceinline__ u32x4 att_grad8(const float* ga, const float* gdl, const float* gdp, const float* gv, unsigned magic, int64_t row, int c, int col8) {
    const unsigned grp = __umulhi((unsigned)row, magic);
    const float aj = ga[row], dj = gdl[row];
    const float* dpq = gdp + (int64_t)grp * c + col8;
    const f32x4 d0 = *(const f32x4*)dpq, d1 = *(const f32x4*)(dpq + 4);
    const f32x4 v0 = *(const f32x4*)(gv + col8), v1 = *(const f32x4*)(gv + col8 + 4);
    u32x4 o;
    o.x = pack2(aj * d0[0] + dj * v0[0], aj * d0[1] + dj * v0[1]);
    o.y = pack2(aj * d0[2] + dj * v0[2], aj * d0[3] + dj * v0[3]);
    o.z = pack2(aj * d1[0] + dj * v1[0], aj * d1[1] + dj * v1[1]);
    o.w = pack2(aj * d1[2] + dj * v1[2], aj * d1[3] + dj * v1[3]);
    return o;
}

// R = 16-row tiles a wave carries through the weights together: 2, except 1 in the input-gradient kernel with 128 channels per wave,
// whose epilogue (x, scale, shift, two sums per channel) would not fit the registers next to two tiles of accumulators
template <int CO, bool DX>
constexpr int tiles_of() { return (DX && CO >= 128) ? 1 : 2; }

template <int CK, int CO, bool DX, int POOL = 0>            // POOL: 0 gradient read from src, 1 pooled_grad8, 2 att_grad8
__global__ __launch_bounds__(NT, 1) void rows_layer_kernel(const LayerArgs a) {
    constexpr int R = tiles_of<CO, DX>();
    constexpr int KS = CK / 32;                     // k-steps of 32 channels
    constexpr int KC = KS < 4 ? KS : 4;             // k-steps whose B fragments are in flight together
    constexpr int NOB = CO / 16;                    // 16-channel output blocks
    constexpr int HALVES = CO > 128 ? 2 : 1;        // a wave carries at most 128 output channels (registers)
    constexpr int CW = CO / HALVES;
    constexpr int NB = CW / 16;
    constexpr int NG = NB / 4;                      // groups of 4 blocks = 64 channels = 16 per lane
    constexpr int SLOTS = NWAVE / HALVES;           // row-tile groups in flight per workgroup

    bf16x8* wimg = (bf16x8*)smem;                                   // [KS][NOB][64] A fragments
    float* pa = (float*)(wimg + KS * NOB * 64);                     // [CK]
    float* pb = pa + CK;                                            // [CK]
    float* ea = pb + CK;                                            // [CO] fwd: bias        dx: scale_in
    float* eb = ea + CO;                                            // [CO]                  dx: shift_in
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int half = wave % HALVES, slot = wave / HALVES;

    // ---- weights -> A fragments.  fwd: A[m <-> cout][k <-> cin] = W[cout][cin];  dx: A[m <-> cin][k <-> cout] = W[cout][cin]
    for (int i = threadIdx.x; i < CO * (CK / 8); i += NT) {
        int co, kc;
        if (!DX) { co = i / (CK / 8); kc = i % (CK / 8); } else { kc = i / CO; co = i % CO; }      // consecutive threads: consecutive fp32 words of W
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = DX ? a.w[(int64_t)(8 * kc + j) * CO + co] : a.w[(int64_t)co * CK + 8 * kc + j];
        int ob, m;
        block_row_of(co, ob, m);
        u32x4 p = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        wimg[((kc >> 2) * NOB + ob) * 64 + (kc & 3) * 16 + m] = as_frag(p);
    }
    const bool has_pre = a.pre_a != nullptr;
    for (int i = threadIdx.x; i < CK; i += NT) {
        pa[i] = has_pre ? a.pre_a[i] : (DX ? 0.f : 1.f);
        pb[i] = has_pre ? a.pre_b[i] : 0.f;
    }
    for (int i = threadIdx.x; i < CO; i += NT) {
        if (!DX) { ea[i] = a.bias ? a.bias[i] : 0.f; eb[i] = 0.f; }
        else { ea[i] = a.in_scale ? a.in_scale[i] : 1.f; eb[i] = a.in_shift ? a.in_shift[i] : 0.f; }
    }
    __syncthreads();

    const float relu_floor = a.pre_relu ? 0.f : -INFINITY;
    const bool bare_relu = !DX && !has_pre && a.pre_relu;        // ReLU without an affine part: integer max on the packed pairs
    const bool dx_y = DX && a.src2 != nullptr;
    const bool dx_x = DX && a.xin != nullptr;
    const float in_floor = a.in_relu ? 0.f : -INFINITY;

    float st0[NB][4], st1[NB][4];                    // per-lane partial sums (fwd: y, y^2;  dx: d * x, d)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st0[b][r] = 0.f; st1[b][r] = 0.f; }

    const int64_t nunits = (a.rows + 16 * R - 1) / (16 * R);
    for (int64_t u = (int64_t)blockIdx.x * SLOTS + slot; u < nunits; u += (int64_t)gridDim.x * SLOTS) {
        const int64_t r0 = u * (16 * R);
        if (KS * NB > 8) asm volatile("" ::: "memory");      // keeps the A fragments in LDS: hoisted out of this loop they would take KS * NB * 4 registers
        f32x4 acc[R][NB];
#pragma unroll
        for (int t = 0; t < R; ++t)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        int64_t rowc[R];
        bool valid[R];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const int64_t row = r0 + 16 * t + n;
            valid[t] = row < a.rows;
            rowc[t] = valid[t] ? row : a.rows - 1;
        }
        // the B fragments of KC k-steps are requested together, then consumed; the other wave of the SIMD computes meanwhile
#pragma unroll 1
        for (int s0 = 0; s0 < KS; s0 += KC) {
            u32x4 raw[KC][R], raw2[KC][R];
#pragma unroll
            for (int sc = 0; sc < KC; ++sc)
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    if constexpr (POOL == 1) raw[sc][t] = pooled_grad8(a.src, a.garg, a.pool_p, a.pool_magic, rowc[t], CK, 32 * (s0 + sc) + 8 * g);
                    else if constexpr (POOL == 2) raw[sc][t] = att_grad8(a.g_a, a.g_dl, a.g_dp, a.g_v, a.pool_magic, rowc[t], CK, 32 * (s0 + sc) + 8 * g);
                    else raw[sc][t] = *(const u32x4*)(a.src + rowc[t] * CK + 32 * (s0 + sc) + 8 * g);
                    if (dx_y) raw2[sc][t] = *(const u32x4*)(a.src2 + rowc[t] * CK + 32 * (s0 + sc) + 8 * g);
                }
#pragma unroll
            for (int sc = 0; sc < KC; ++sc) {
                const int s = s0 + sc;
                bf16x8 bfr[R];
                if ((!DX && has_pre) || dx_y) {
                    const f32x4 a0 = *(const f32x4*)(pa + 32 * s + 8 * g), a1 = *(const f32x4*)(pa + 32 * s + 8 * g + 4);
                    const f32x4 b0 = *(const f32x4*)(pb + 32 * s + 8 * g), b1 = *(const f32x4*)(pb + 32 * s + 8 * g + 4);
#pragma unroll
                    for (int t = 0; t < R; ++t) {
                        const u32x4 q = raw[sc][t];
                        float e[8] = {lo16(q.x), hi16(q.x), lo16(q.y), hi16(q.y), lo16(q.z), hi16(q.z), lo16(q.w), hi16(q.w)};
                        if (!DX) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                e[j] = fmaxf(__builtin_fmaf(e[j], a0[j], b0[j]), relu_floor);
                                e[4 + j] = fmaxf(__builtin_fmaf(e[4 + j], a1[j], b1[j]), relu_floor);
                            }
                        } else {
                            const u32x4 q2 = raw2[sc][t];
                            const float y[8] = {lo16(q2.x), hi16(q2.x), lo16(q2.y), hi16(q2.y), lo16(q2.z), hi16(q2.z), lo16(q2.w), hi16(q2.w)};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                e[j] = __builtin_fmaf(y[j], b0[j], e[j] + a0[j]);
                                e[4 + j] = __builtin_fmaf(y[4 + j], b1[j], e[4 + j] + a1[j]);
                            }
                        }
                        const u32x4 p = {pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7])};
                        bfr[t] = as_frag(p);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < R; ++t) bfr[t] = as_frag(bare_relu ? relu_bf16x8(raw[sc][t]) : raw[sc][t]);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const bf16x8 af = wimg[(s * NOB + half * NB + b) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < R; ++t) acc[t][b] = PPS_MFMA16(af, bfr[t], acc[t][b], 0, 0, 0);
                }
            }
        }
        // ---- epilogue: lane (row n, g) holds channels half*CW + 64 grp + 16 g + [0, 16)
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const int64_t row = rowc[t];
            const float live = valid[t] ? 1.f : 0.f;
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
                const int c0 = half * CW + 64 * grp + 16 * g;
                unsigned out[8];
                if (!DX) {
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const f32x4 bi = *(const f32x4*)(ea + c0 + 4 * o);
                        const f32x4 v = acc[t][4 * grp + o] + bi;
                        out[2 * o] = pack2(v[0], v[1]);
                        out[2 * o + 1] = pack2(v[2], v[3]);
                        if (a.partials) {
                            const float y4[4] = {lo16(out[2 * o]), hi16(out[2 * o]), lo16(out[2 * o + 1]), hi16(out[2 * o + 1])};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                st0[4 * grp + o][r] += live * y4[r];
                                st1[4 * grp + o][r] += live * y4[r] * y4[r];
                            }
                        }
                    }
                } else {
                    u32x4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
                    if (dx_x) {
                        x0 = *(const u32x4*)(a.xin + row * CO + c0);
                        x1 = *(const u32x4*)(a.xin + row * CO + c0 + 8);
                    }
                    const unsigned xw[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    u32x4 g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
                    if (a.addend) {
                        g0 = *(const u32x4*)(a.addend + row * CO + c0);
                        g1 = *(const u32x4*)(a.addend + row * CO + c0 + 8);
                    }
                    float att = 0.f;
                    if (a.att_a) {                    // (host: never together with addend)
                        const uint16_t* dpq = a.att_dp + (int64_t)(a.att_k == 1 ? (unsigned)row : __umulhi((unsigned)row, a.att_magic)) * CO + c0;
                        att = a.att_a[row];
                        g0 = *(const u32x4*)dpq;
                        g1 = *(const u32x4*)(dpq + 8);
                    }
                    const unsigned gw[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const f32x4 sc = *(const f32x4*)(ea + c0 + 4 * o), sh = *(const f32x4*)(eb + c0 + 4 * o);
                        const float x4[4] = {lo16(xw[2 * o]), hi16(xw[2 * o]), lo16(xw[2 * o + 1]), hi16(xw[2 * o + 1])};
                        float d[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pre = __builtin_fmaf(x4[r], sc[r], sh[r]);
                            d[r] = (!dx_x || pre > in_floor) ? acc[t][4 * grp + o][r] : 0.f;          // ReLU mask of the layer's input
                            st0[4 * grp + o][r] += live * d[r] * x4[r];
                            st1[4 * grp + o][r] += live * d[r];
                            d[r] *= sc[r];
                        }
                        const float ad[4] = {lo16(gw[2 * o]), hi16(gw[2 * o]), lo16(gw[2 * o + 1]), hi16(gw[2 * o + 1])};
                        if (a.att_a) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) d[r] += x4[r] > 0.f ? att * ad[r] : 0.f;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) d[r] += ad[r];
                        }
                        out[2 * o] = pack2(d[0], d[1]);
                        out[2 * o + 1] = pack2(d[2], d[3]);
                    }
                }
                if (valid[t]) {
                    uint16_t* dstp = a.dst + row * CO + c0;
                    *(u32x4*)dstp = u32x4{out[0], out[1], out[2], out[3]};
                    *(u32x4*)(dstp + 8) = u32x4{out[4], out[5], out[6], out[7]};
                }
            }
        }
    }

    // ---- per-channel sums of the workgroup, in a fixed order: lanes of a row group (shuffles), then the SLOTS waves of a half (LDS)
    if (a.partials) {
        __syncthreads();                              // the weight image is no longer needed: reuse it
        float* red = (float*)smem;                    // [SLOTS][2][CO]
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s0 = row16_sum(st0[b][r]), s1 = row16_sum(st1[b][r]);
                if (n == 0) {
                    const int c = half * CW + 64 * (b >> 2) + 16 * g + 4 * (b & 3) + r;
                    red[(slot * 2 + 0) * CO + c] = s0;
                    red[(slot * 2 + 1) * CO + c] = s1;
                }
            }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * CO; i += NT) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) s += red[q * 2 * CO + i];
            a.partials[(int64_t)blockIdx.x * 2 * CO + i] = s;
        }
    }
}

template <int CK, int CO>
constexpr size_t layer_lds() {
    return (size_t)CK * CO * 2 + (size_t)(2 * CK + 2 * CO) * 4;
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient: dW[cout][cin] = sum_rows G[row][cout] * act(x)[row][cin],  db[cout] = sum_rows G[row][cout]
// ---------------------------------------------------------------------------------------------------------------------
struct DwArgs {
    const uint16_t* gy;           // [rows, CO]
    const uint16_t* y;            // [rows, CO] or NULL
    const float* gs;              // [CO] or NULL
    const float* gq2;             // [CO] or NULL
    const uint16_t* x;            // [rows, CI]
    const float* in_scale;        // [CI] or NULL
    const float* in_shift;
    int in_relu;
    float* dw_part;               // [gridDim.x][CO][CI]
    float* db_part;               // [gridDim.x][CO]
    int64_t rows;
    const uint8_t* garg;          // POOL: gy is the gradient per group [rows / pool_p, CO], garg the winning rows (see pooled_grad8)
    int pool_p;
    unsigned pool_magic;
    const float* g_a;             // POOL == 2: gy is not read, see LayerArgs / att_grad8
    const float* g_dl;
    const float* g_dp;
    const float* g_v;
};

__device__ __forceinline__ u32x2 lds_tr_read(const uint16_t* p) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)p) : "memory");
    return v;
}
// the compiler does not count the reads above: every fragment passes through a wait before its first use (the first one waits, the rest are free)
__device__ __forceinline__ void lds_tr_wait(u32x2& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory"); }

template <int CI, int CO>
constexpr int dw_ksteps() { return CI + CO <= 128 ? 4 : (CI + CO <= 384 ? 2 : 1); }

template <int CI, int CO, bool HAS_Y, int POOL = 0>
__global__ __launch_bounds__(NT, 1) void rows_dw_kernel(const DwArgs a) {
    constexpr int KRS = dw_ksteps<CI, CO>();        // MFMA contraction steps (32 rows each) per staged tile: narrow layers stage more rows per barrier
    constexpr int KR = 32 * KRS;                    // rows per step
    constexpr int PG = CO + 16, PA = CI + 16;       // LDS row pitch in elements: + 32 bytes keeps the transpose reads of a half-wave on 64 banks
    constexpr int WM = 4, WN = 2;                   // waves along cout / cin
    constexpr int MB = CO / 16 / WM, NBK = CI / 16 / WN;
    constexpr int GCH = KR * CO / 8, ACH = KR * CI / 8;                     // 16-byte chunks of a tile
    constexpr int GIT = (GCH + NT - 1) / NT, AIT = (ACH + NT - 1) / NT;

    uint16_t* gimg = (uint16_t*)smem;                                       // [2][KR][PG]
    uint16_t* aimg = gimg + 2 * KR * PG;                                    // [2][KR][PA]
    float* prm = (float*)(aimg + 2 * KR * PA);                              // gs[CO], gq2[CO], scale[CI], shift[CI]
    float* gs = prm, *gq2 = prm + CO, *isc = prm + 2 * CO, *ish = prm + 2 * CO + CI;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i16 = lane & 15, kg = lane >> 4;
    const int wm = wave % WM, wn = wave / WM;
    constexpr bool has_y = HAS_Y;                                           // a.y != NULL: the statistics of y carry gradient (a BatchNorm follows)
    const bool has_aff = a.in_scale != nullptr;
    // per-channel parameters, stored [j][chunk] (channel 8 chunk + j): the threads of a wave hold consecutive chunks, so reading element j of
    // every chunk is conflict-free (channel-major storage is an 8-way bank conflict per read -- it doubled the kernel's time)
    for (int i = threadIdx.x; i < CO; i += NT) {
        const int t = (i & 7) * (CO / 8) + (i >> 3);
        gs[t] = has_y ? a.gs[i] : 0.f;
        gq2[t] = has_y ? a.gq2[i] : 0.f;
    }
    for (int i = threadIdx.x; i < CI; i += NT) {
        const int t = (i & 7) * (CI / 8) + (i >> 3);
        isc[t] = has_aff ? a.in_scale[i] : 1.f;
        ish[t] = has_aff ? a.in_shift[i] : 0.f;
    }
    __syncthreads();
    const float in_floor = a.in_relu ? 0.f : -INFINITY;
    const bool relu_only = !has_aff && a.in_relu;                 // a bare ReLU on the input: integer max on the packed bf16 pairs

    // slab of rows of this workgroup: whole steps of KR rows
    const int64_t nsteps = (a.rows + KR - 1) / KR;
    const int64_t per = (nsteps + gridDim.x - 1) / gridDim.x;
    const int64_t s_begin = (int64_t)blockIdx.x * per, s_end = (s_begin + per < nsteps) ? s_begin + per : nsteps;

    f32x4 acc[MB][NBK];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int k = 0; k < NBK; ++k) acc[m][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) db[j] = 0.f;

    struct Regs { u32x4 g[GIT], y[HAS_Y ? GIT : 1], x[AIT]; };
    auto fetch = [&](Regs& rr_, int64_t step) {
        u32x4 (&rg)[GIT] = rr_.g; u32x4 (&ry)[HAS_Y ? GIT : 1] = rr_.y; u32x4 (&rx)[AIT] = rr_.x;
        const int64_t r0 = step * KR;
#pragma unroll
        for (int it = 0; it < GIT; ++it) {
            const int id = threadIdx.x + NT * it;
            const int row = id / (CO / 8), ch = id % (CO / 8);
            const int64_t rr = r0 + row < a.rows ? r0 + row : a.rows - 1;     // rows past the end re-read the last row (zeroed when staged): no
            if (id < GCH) {                                                  // lane-dependent branch around the loads, they all issue back to back
                if constexpr (POOL == 1) rg[it] = pooled_grad8(a.gy, a.garg, a.pool_p, a.pool_magic, rr, CO, 8 * ch);
                // (POOL == 2, tried: the ingredients -- 2 floats of the row, 8 of its group -- carried in the registers and put together in stage(),
                // so that no arithmetic waits for loads in front of the MFMA section: 107 spilled VGPRs, the kernel has 4 registers to spare)
                else if constexpr (POOL == 2) rg[it] = att_grad8(a.g_a, a.g_dl, a.g_dp, a.g_v, a.pool_magic, rr, CO, 8 * ch);
                else rg[it] = *(const u32x4*)(a.gy + rr * CO + 8 * ch);
                if constexpr (HAS_Y) ry[it] = *(const u32x4*)(a.y + rr * CO + 8 * ch);
            }
        }
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int id = threadIdx.x + NT * it;
            const int row = id / (CI / 8), ch = id % (CI / 8);
            const int64_t rr = r0 + row < a.rows ? r0 + row : a.rows - 1;
            if (id < ACH) rx[it] = *(const u32x4*)(a.x + rr * CI + 8 * ch);
        }
    };
    auto stage = [&](const Regs& rr_, int buf, int64_t step) {
        const u32x4 (&rg)[GIT] = rr_.g; const u32x4 (&ry)[HAS_Y ? GIT : 1] = rr_.y; const u32x4 (&rx)[AIT] = rr_.x;
        const int64_t r0 = step * KR;
        uint16_t* gdst = gimg + buf * KR * PG;
        uint16_t* adst = aimg + buf * KR * PA;
#pragma unroll
        for (int it = 0; it < GIT; ++it) {
            const int id = threadIdx.x + NT * it;
            if (id >= GCH) break;
            const int row = id / (CO / 8), ch = id % (CO / 8);
            const bool live = r0 + row < a.rows;
            const u32x4 q = live ? rg[it] : u32x4{0, 0, 0, 0};
            float e[8] = {lo16(q.x), hi16(q.x), lo16(q.y), hi16(q.y), lo16(q.z), hi16(q.z), lo16(q.w), hi16(q.w)};
            u32x4 p = q;
            if constexpr (HAS_Y) {
                const u32x4 q2 = ry[it];
                const float yv[8] = {lo16(q2.x), hi16(q2.x), lo16(q2.y), hi16(q2.y), lo16(q2.z), hi16(q2.z), lo16(q2.w), hi16(q2.w)};
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = live ? __builtin_fmaf(yv[j], gq2[j * (CO / 8) + ch], e[j] + gs[j * (CO / 8) + ch]) : 0.f;
                p = u32x4{pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7])};
                e[0] = lo16(p.x); e[1] = hi16(p.x); e[2] = lo16(p.y); e[3] = hi16(p.y); e[4] = lo16(p.z); e[5] = hi16(p.z); e[6] = lo16(p.w); e[7] = hi16(p.w);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) db[j] += e[j];
            *(u32x4*)(gdst + row * PG + 8 * ch) = p;
        }
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int id = threadIdx.x + NT * it;
            if (id >= ACH) break;
            const int row = id / (CI / 8), ch = id % (CI / 8);
            const bool live = r0 + row < a.rows;
            u32x4 p = live ? rx[it] : u32x4{0, 0, 0, 0};
            if (relu_only) p = relu_bf16x8(p);
            if (has_aff) {
                float e[8] = {lo16(p.x), hi16(p.x), lo16(p.y), hi16(p.y), lo16(p.z), hi16(p.z), lo16(p.w), hi16(p.w)};
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = live ? fmaxf(__builtin_fmaf(e[j], isc[j * (CI / 8) + ch], ish[j * (CI / 8) + ch]), in_floor) : 0.f;
                p = u32x4{pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7])};
            }
            *(u32x4*)(adst + row * PA + 8 * ch) = p;
        }
    };

    auto products = [&](int buf) {
        const uint16_t* gbase = gimg + buf * KR * PG;
        const uint16_t* abase = aimg + buf * KR * PA;
        // operand element j of lane (column i16, kg) = tile row (j < 4 ? 4 kg + j : 16 + 4 kg + j - 4): two transpose reads; lane i16 supplies
        // the 8-byte piece (row 4 kg + (i16 >> 2), columns 4 (i16 & 3) ..) of the 16-column block and receives column i16
        const int prow = 4 * kg + (i16 >> 2), pcol = 4 * (i16 & 3);
        constexpr int KH = NBK > 4 ? 4 : NBK;                       // B fragments are read 4 at a time (registers)
#pragma unroll
        for (int ks = 0; ks < KRS; ++ks) {
        const uint16_t* gsrc = gbase + 32 * ks * PG;
        const uint16_t* asrc = abase + 32 * ks * PA;
        u32x2 af[MB][2];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const uint16_t* p = gsrc + prow * PG + 16 * (wm * MB + m) + pcol;
            af[m][0] = lds_tr_read(p);
            af[m][1] = lds_tr_read(p + 16 * PG);
        }
#pragma unroll
        for (int k0 = 0; k0 < NBK; k0 += KH) {
            u32x2 bf[KH][2];
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const uint16_t* p = asrc + prow * PA + 16 * (wn * NBK + k0 + k) + pcol;
                bf[k][0] = lds_tr_read(p);
                bf[k][1] = lds_tr_read(p + 16 * PA);
            }
            if (k0 == 0) {
#pragma unroll
                for (int m = 0; m < MB; ++m) { lds_tr_wait(af[m][0]); lds_tr_wait(af[m][1]); }
            }
#pragma unroll
            for (int k = 0; k < KH; ++k) { lds_tr_wait(bf[k][0]); lds_tr_wait(bf[k][1]); }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const u32x4 am = {af[m][0].x, af[m][0].y, af[m][1].x, af[m][1].y};
#pragma unroll
                for (int k = 0; k < KH; ++k) {
                    const u32x4 bk = {bf[k][0].x, bf[k][0].y, bf[k][1].x, bf[k][1].y};
                    acc[m][k0 + k] = PPS_MFMA16(as_frag(am), as_frag(bk), acc[m][k0 + k], 0, 0, 0);
                }
            }
        }
        }
    };
    // tiles are fetched TWO steps ahead (two register sets): one 32-row tile in flight per workgroup leaves the memory pipe idle most of the time
    Regs ra, rb;
    if (s_begin < s_end) fetch(ra, s_begin);
    if (s_begin + 1 < s_end) fetch(rb, s_begin + 1);
    for (int64_t step = s_begin; step < s_end; step += 2) {
        stage(ra, 0, step);
        if (step + 2 < s_end) fetch(ra, step + 2);
        __syncthreads();                                            // also orders: the products of the previous step (other buffer) precede the next stage into it
        products(0);
        if (step + 1 < s_end) {
            stage(rb, 1, step + 1);
            if (step + 3 < s_end) fetch(rb, step + 3);
            __syncthreads();
            products(1);
        }
    }

    // ---- slab partials: D row 4 g + r <-> cout, column i16 <-> cin
    float* dwp = a.dw_part + (int64_t)blockIdx.x * CO * CI;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int k = 0; k < NBK; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) dwp[(int64_t)(16 * (wm * MB + m) + 4 * kg + r) * CI + 16 * (wn * NBK + k) + i16] = acc[m][k][r];
    if (a.db_part) {
        // a thread always stages the same 8 channels (NT is a multiple of CO / 8): threads with equal chunk are summed in thread order
        __syncthreads();
        float* red = (float*)smem;                                   // [NT][8]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = db[j];
        __syncthreads();
        for (int c = threadIdx.x; c < CO; c += NT) {
            const int ch = c >> 3, j = c & 7;
            float s = 0.f;
            for (int t = ch; t < NT && t < GCH; t += CO / 8) s += red[t * 8 + j];
            a.db_part[(int64_t)blockIdx.x * CO + c] = s;
        }
    }
}

template <int CI, int CO>
constexpr size_t dw_lds() {
    return (size_t)2 * 32 * dw_ksteps<CI, CO>() * (CO + 16 + CI + 16) * 2 + (size_t)(2 * CO + 2 * CI) * 4;
}

// out[i] = sum_p part[p][i]: SL threads share an element (p = slice, slice + SL, ... in double), their sums are added in slice order
template <int SL>
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int np, int64_t n, float* __restrict__ out) {
    constexpr int E = 256 / SL;
    __shared__ double red[SL][E];
    const int e = threadIdx.x % E, sl = threadIdx.x / E;
    const int64_t i = (int64_t)blockIdx.x * E + e;
    double s = 0.0;
    if (i < n) {
#pragma unroll 8
        for (int p = sl; p < np; p += SL) s += (double)part[(int64_t)p * n + i];
    }
    red[sl][e] = s;
    __syncthreads();
    if (sl == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < SL; ++q) t += red[q][e];
        out[i] = (float)t;
    }
}

void launch_sum_partials(const float* part, int np, int64_t n, float* out, hipStream_t st) {
    if (n <= 4096) hipLaunchKernelGGL(sum_partials_kernel<16>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, part, np, n, out);
    else hipLaunchKernelGGL(sum_partials_kernel<4>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, part, np, n, out);
}

// BatchNorm parameters of the output from the slab partials (sum y, sum y^2): scale = gamma rstd, shift = beta - mean scale;
// save = (mean, rstd); running statistics like torch.nn.BatchNorm1d (biased variance for the batch, unbiased for the running value).
// 16 channels per workgroup, 16 threads per channel over the partials.
__global__ __launch_bounds__(256) void bn_affine_kernel(const float* __restrict__ part, int np, int c, double rows, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, float momentum, float eps, float* __restrict__ affine,
                                                       float* __restrict__ save) {
    __shared__ double red[2][16][16];
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + e;
    double s = 0.0, q = 0.0;
    if (i < c) {
#pragma unroll 4
        for (int p = sl; p < np; p += 16) {
            s += (double)part[(int64_t)p * 2 * c + i];
            q += (double)part[(int64_t)p * 2 * c + c + i];
        }
    }
    red[0][sl][e] = s;
    red[1][sl][e] = q;
    __syncthreads();
    if (sl != 0 || i >= c) return;
    s = 0.0; q = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) { s += red[0][t][e]; q += red[1][t][e]; }
    const double mean = s / rows;
    double var = q / rows - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma[i] * rstd;
    affine[i] = (float)sc;
    affine[c + i] = (float)((double)beta[i] - mean * sc);
    save[i] = (float)mean;
    save[c + i] = (float)rstd;
    if (running_mean && running_var) {
        const double unbiased = rows > 1.0 ? var * rows / (rows - 1.0) : var;
        running_mean[i] = (float)((1.0 - momentum) * (double)running_mean[i] + momentum * mean);
        running_var[i] = (float)((1.0 - momentum) * (double)running_var[i] + momentum * unbiased);
    }
}

// gradients wrt (scale, shift) of the output -> what every row of y receives (gS, 2 gQ) + dgamma, dbeta
__global__ void bn_affine_bwd_kernel(const float* __restrict__ d_affine, const float* __restrict__ save, const float* __restrict__ gamma, int c,
                                     double rows, float* __restrict__ gstat, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const double ds = d_affine[i], dt = d_affine[c + i], mean = save[i], rstd = save[c + i], gm = gamma[i];
    const double dsc = ds - mean * dt;                       // total gradient of scale = gamma * rstd (shift = beta - mean * scale)
    dgamma[i] = (float)(rstd * dsc);
    dbeta[i] = (float)dt;
    const double drstd = gm * dsc;
    const double dvar = -0.5 * rstd * rstd * rstd * drstd;
    const double dmean = -dt * gm * rstd - 2.0 * mean * dvar;  // var = Q / rows - mean^2
    gstat[i] = (float)(dmean / rows);
    gstat[c + i] = (float)(2.0 * dvar / rows);
}

// per (group, channel) extrema of raw rows: x [groups, p, c] bf16 -> mx, mn [groups, c] fp32 and the row (0..p-1) of each (first occurrence).
// A monotone per-channel activation commutes with the extremum: max_p relu(x s + t) = relu(s * (s >= 0 ? max_p x : min_p x) + t), so the
// max-pool over the patch (source/base/nn.py:181) needs only these, not the activated tensor.  One wave per group, 4 channels per lane.
__global__ __launch_bounds__(256) void rows_extrema_kernel(const uint16_t* __restrict__ x, int64_t groups, int p, int c, float* __restrict__ mx,
                                                          float* __restrict__ mn, int* __restrict__ amx, int* __restrict__ amn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t q = (int64_t)blockIdx.x * 4 + wave; q < groups; q += (int64_t)gridDim.x * 4) {
        for (int c0 = 4 * lane; c0 < c; c0 += 256) {
            float hi[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, lo[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
            int ihi[4] = {0, 0, 0, 0}, ilo[4] = {0, 0, 0, 0};
            const uint16_t* src = x + q * (int64_t)p * c + c0;
            for (int j = 0; j < p; ++j) {
                const u32x2 u = *(const u32x2*)(src + (int64_t)j * c);
                const float v[4] = {lo16(u.x), hi16(u.x), lo16(u.y), hi16(u.y)};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (v[r] > hi[r]) { hi[r] = v[r]; ihi[r] = j; }
                    if (v[r] < lo[r]) { lo[r] = v[r]; ilo[r] = j; }
                }
            }
            *(f32x4*)(mx + q * c + c0) = f32x4{hi[0], hi[1], hi[2], hi[3]};
            *(f32x4*)(mn + q * c + c0) = f32x4{lo[0], lo[1], lo[2], lo[3]};
            *(int4*)(amx + q * c + c0) = make_int4(ihi[0], ihi[1], ihi[2], ihi[3]);
            *(int4*)(amn + q * c + c0) = make_int4(ilo[0], ilo[1], ilo[2], ilo[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// First layer of PointNet: 3 coordinates -> 64 channels (conv0a, source/base/nn.py:323) -- not an MFMA shape.  A thread computes 8 channels
// of a row (16-byte store), 8 threads a row; statistics / weight gradients per thread, reduced over the block in a fixed order.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int R3_ROWS = 32;            // rows per block iteration (256 threads / 8 channel chunks)

__global__ __launch_bounds__(256) void rows3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       int64_t rows, uint16_t* __restrict__ y, float* __restrict__ partials) {
    __shared__ float red[256][17];
    const int ch = threadIdx.x & 7, rl = threadIdx.x >> 3;
    float wv[8][3], bv[8], s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        wv[j][0] = w[(8 * ch + j) * 3]; wv[j][1] = w[(8 * ch + j) * 3 + 1]; wv[j][2] = w[(8 * ch + j) * 3 + 2];
        bv[j] = bias ? bias[8 * ch + j] : 0.f;
        s0[j] = 0.f; s1[j] = 0.f;
    }
    for (int64_t row = (int64_t)blockIdx.x * R3_ROWS + rl; row < rows; row += (int64_t)gridDim.x * R3_ROWS) {
        const float x0 = x[row * 3], x1 = x[row * 3 + 1], x2 = x[row * 3 + 2];
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(wv[j][2], x2, __builtin_fmaf(wv[j][1], x1, __builtin_fmaf(wv[j][0], x0, bv[j])));
        const u32x4 p = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
        *(u32x4*)(y + row * 64 + 8 * ch) = p;
        const float r[8] = {lo16(p.x), hi16(p.x), lo16(p.y), hi16(p.y), lo16(p.z), hi16(p.z), lo16(p.w), hi16(p.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0[j] += r[j]; s1[j] += r[j] * r[j]; }
    }
    if (!partials) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x][j] = s0[j]; red[threadIdx.x][8 + j] = s1[j]; }
    __syncthreads();
    if (threadIdx.x < 128) {                                      // (statistic, channel): threads of equal chunk in row-lane order
        const int st = threadIdx.x >> 6, c = threadIdx.x & 63;
        float t = 0.f;
        for (int q = 0; q < R3_ROWS; ++q) t += red[q * 8 + (c >> 3)][8 * st + (c & 7)];
        partials[(int64_t)blockIdx.x * 128 + st * 64 + c] = t;
    }
}

// partial [block][256]: dW [64][3] then db [64]
__global__ __launch_bounds__(256) void rows3_bwd_kernel(const float* __restrict__ x, const uint16_t* __restrict__ y, const uint16_t* __restrict__ gy,
                                                       const float* __restrict__ gstat, int64_t rows, float* __restrict__ partials) {
    __shared__ float red[256][33];
    const int ch = threadIdx.x & 7, rl = threadIdx.x >> 3;
    float gs[8], gq[8], acc[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        gs[j] = gstat ? gstat[8 * ch + j] : 0.f;
        gq[j] = gstat ? gstat[64 + 8 * ch + j] : 0.f;
        acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.f;
    }
    for (int64_t row = (int64_t)blockIdx.x * R3_ROWS + rl; row < rows; row += (int64_t)gridDim.x * R3_ROWS) {
        const float x0 = x[row * 3], x1 = x[row * 3 + 1], x2 = x[row * 3 + 2];
        const u32x4 g4 = *(const u32x4*)(gy + row * 64 + 8 * ch);
        float g[8] = {lo16(g4.x), hi16(g4.x), lo16(g4.y), hi16(g4.y), lo16(g4.z), hi16(g4.z), lo16(g4.w), hi16(g4.w)};
        if (gstat) {
            const u32x4 y4 = *(const u32x4*)(y + row * 64 + 8 * ch);
            const float yv[8] = {lo16(y4.x), hi16(y4.x), lo16(y4.y), hi16(y4.y), lo16(y4.z), hi16(y4.z), lo16(y4.w), hi16(y4.w)};
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = __builtin_fmaf(yv[j], gq[j], g[j] + gs[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[j][0] += g[j] * x0; acc[j][1] += g[j] * x1; acc[j][2] += g[j] * x2; acc[j][3] += g[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[threadIdx.x][4 * j + k] = acc[j][k];
    __syncthreads();
    {
        const int c = threadIdx.x >> 2, k = threadIdx.x & 3;       // 64 channels x (3 weights + bias)
        float t = 0.f;
        for (int q = 0; q < R3_ROWS; ++q) t += red[q * 8 + (c >> 3)][4 * (c & 7) + k];
        float* out = partials + (int64_t)blockIdx.x * 256;
        if (k < 3) out[c * 3 + k] = t; else out[192 + c] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Feature transform of PointNet (source/base/nn.py:330-331, torch.bmm(trans2, x)): per group q of p <= 64 rows
//     out[q, i, :] = act(x)[q, i, :] T[q]^T,      x [Q*p, 64] bf16 stored activation, T [Q, 64, 64] bf16 (+ I), out [Q*p, 64] bf16
// forward: a wave per group, T[q] and the rows straight from global memory as MFMA operands.
// backward: a workgroup per group; T[q], act(x) and the output gradient G go row-major into LDS and the three products
//     dA = G T (-> dx, d scale, d shift),    dT^T-free form  dT[j][k] = sum_i G[i][j] act(x)[i][k]
// take their column operands with ds_read_b64_tr_b16.
// ---------------------------------------------------------------------------------------------------------------------
struct PtArgs {
    const uint16_t* x;            // [Q*p, 64]
    const float* scale;           // [64] or NULL
    const float* shift;
    int relu;
    const uint16_t* t;            // [Q, 64, 64]
    int add_identity;
    int64_t nq;
    int p;
    uint16_t* out;                // fwd: [Q*p, 64]
    const uint16_t* g;            // bwd: d out [Q*p, 64]
    uint16_t* dx;                 // bwd: [Q*p, 64]
    uint16_t* dt;                 // bwd: [Q, 64, 64]
    float* partials;              // bwd: [gridDim.x][2][64] (d scale, d shift) or NULL
};

__device__ __forceinline__ u32x4 act8(const u32x4 q, const float* sc, const float* sh, float floor_) {
    float e[8] = {lo16(q.x), hi16(q.x), lo16(q.y), hi16(q.y), lo16(q.z), hi16(q.z), lo16(q.w), hi16(q.w)};
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = fmaxf(__builtin_fmaf(e[j], sc[j], sh[j]), floor_);
    return u32x4{pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7])};
}

// 8 consecutive elements of row j of T starting at column k0, + 1 on the diagonal
__device__ __forceinline__ u32x4 t_chunk(const uint16_t* __restrict__ tq, int j, int k0, int add_identity) {
    u32x4 v = *(const u32x4*)(tq + j * 64 + k0);
    if (add_identity && j >= k0 && j < k0 + 8) {
        unsigned* w = (unsigned*)&v;
        const int e = j - k0;
        const float f = (e & 1) ? hi16(w[e >> 1]) : lo16(w[e >> 1]);
        const unsigned r = pack2(f + 1.f, 0.f) & 0xffffu;
        w[e >> 1] = (e & 1) ? ((w[e >> 1] & 0xffffu) | (r << 16)) : ((w[e >> 1] & 0xffff0000u) | r);
    }
    return v;
}

__global__ __launch_bounds__(256, 2) void patch_transform_fwd_kernel(const PtArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const bool has_act = a.scale != nullptr;
    const float floor_ = a.relu ? 0.f : -INFINITY;
    float sc[2][8], sh[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[s][j] = has_act ? a.scale[32 * s + 8 * g + j] : 1.f;
            sh[s][j] = has_act ? a.shift[32 * s + 8 * g + j] : 0.f;
        }
    for (int64_t q = (int64_t)blockIdx.x * 4 + wave; q < a.nq; q += (int64_t)gridDim.x * 4) {
        const uint16_t* tq = a.t + q * 4096;
        const uint16_t* xq = a.x + q * (int64_t)a.p * 64;
        bf16x8 af[2][4], bfr[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)                        // MFMA row m <-> output channel 16 (m >> 2) + 4 ob + (m & 3): 16 consecutive per lane in the result
                af[s][ob] = as_frag(t_chunk(tq, 16 * (n >> 2) + 4 * ob + (n & 3), 32 * s + 8 * g, a.add_identity));
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = 16 * t + n < a.p ? 16 * t + n : a.p - 1;
                const u32x4 raw = *(const u32x4*)(xq + row * 64 + 32 * s + 8 * g);
                bfr[s][t] = as_frag((has_act || a.relu) ? act8(raw, sc[s], sh[s], floor_) : raw);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc[4];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) acc[ob] = PPS_MFMA16(af[s][ob], bfr[s][t], acc[ob], 0, 0, 0);
            }
            if (16 * t + n < a.p) {
                uint16_t* dst = a.out + (q * a.p + 16 * t + n) * 64 + 16 * g;
                *(u32x4*)dst = u32x4{pack2(acc[0][0], acc[0][1]), pack2(acc[0][2], acc[0][3]), pack2(acc[1][0], acc[1][1]), pack2(acc[1][2], acc[1][3])};
                *(u32x4*)(dst + 8) = u32x4{pack2(acc[2][0], acc[2][1]), pack2(acc[2][2], acc[2][3]), pack2(acc[3][0], acc[3][1]), pack2(acc[3][2], acc[3][3])};
            }
        }
    }
}

__device__ __forceinline__ u32x2 lds_tr_read(const uint16_t* p);
__device__ __forceinline__ void lds_tr_wait(u32x2& v);

constexpr int PT_PITCH = 80;            // LDS row pitch in elements (64 + 16)

__global__ __launch_bounds__(256, 2) void patch_transform_bwd_kernel(const PtArgs a) {
    __shared__ __attribute__((aligned(16))) uint16_t timg[64 * PT_PITCH], ximg[64 * PT_PITCH], gimg[64 * PT_PITCH];
    __shared__ float scl[64], shl[64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i16 = lane & 15, kg = lane >> 4;
    const bool has_act = a.scale != nullptr;
    const float floor_ = a.relu ? 0.f : -INFINITY;
    if (threadIdx.x < 64) { scl[threadIdx.x] = has_act ? a.scale[threadIdx.x] : 1.f; shl[threadIdx.x] = has_act ? a.shift[threadIdx.x] : 0.f; }
    __syncthreads();
    float dsc[4] = {0.f, 0.f, 0.f, 0.f}, dsh[4] = {0.f, 0.f, 0.f, 0.f};       // channels 16 w + 4 kg + r
    const f32x4 sc4 = *(const f32x4*)(scl + 16 * w + 4 * kg), sh4 = *(const f32x4*)(shl + 16 * w + 4 * kg);

    for (int64_t q = blockIdx.x; q < a.nq; q += gridDim.x) {
        const uint16_t* tq = a.t + q * 4096;
        const uint16_t* xq = a.x + q * (int64_t)a.p * 64;
        const uint16_t* gq = a.g + q * (int64_t)a.p * 64;
        __syncthreads();                                          // the previous group's operands have been read
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int id = threadIdx.x + 256 * it, row = id >> 3, ch = id & 7;
            *(u32x4*)(timg + row * PT_PITCH + 8 * ch) = t_chunk(tq, row, 8 * ch, a.add_identity);
            u32x4 xv = {0, 0, 0, 0}, gv = {0, 0, 0, 0};
            if (row < a.p) {
                xv = *(const u32x4*)(xq + row * 64 + 8 * ch);
                gv = *(const u32x4*)(gq + row * 64 + 8 * ch);
                if (has_act || a.relu) xv = act8(xv, scl + 8 * ch, shl + 8 * ch, floor_);
            }
            *(u32x4*)(ximg + row * PT_PITCH + 8 * ch) = xv;       // rows beyond p: zeros in both operands
            *(u32x4*)(gimg + row * PT_PITCH + 8 * ch) = gv;
        }
        __syncthreads();
        const int prow = 4 * kg + (i16 >> 2), pcol = 4 * (i16 & 3);

        // ---- dA[i][k] = sum_j G[i][j] T[j][k]:  D[m <-> k = 16 w + m][n <-> row i],  contraction over j in two steps of 32
        f32x4 da[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) da[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x2 a0 = lds_tr_read(timg + (32 * s + prow) * PT_PITCH + 16 * w + pcol);
            u32x2 a1 = lds_tr_read(timg + (32 * s + 16 + prow) * PT_PITCH + 16 * w + pcol);
            lds_tr_wait(a0); lds_tr_wait(a1);
            const u32x4 am = {a0.x, a0.y, a1.x, a1.y};            // element e < 4: j = 32 s + 4 kg + e;  e >= 4: j = 32 s + 16 + 4 kg + e - 4
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint16_t* gr = gimg + (16 * t + i16) * PT_PITCH + 32 * s + 4 * kg;
                const u32x2 b0 = *(const u32x2*)gr, b1 = *(const u32x2*)(gr + 16);
                const u32x4 bm = {b0.x, b0.y, b1.x, b1.y};
                da[t] = PPS_MFMA16(as_frag(am), as_frag(bm), da[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 16 * t + i16;
            if (row < a.p) {
                const int64_t off = (q * a.p + row) * 64 + 16 * w + 4 * kg;
                const u32x2 xr = *(const u32x2*)(a.x + off);
                const float x4[4] = {lo16(xr.x), hi16(xr.x), lo16(xr.y), hi16(xr.y)};
                float d[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pre = __builtin_fmaf(x4[r], sc4[r], sh4[r]);
                    d[r] = pre > floor_ ? da[t][r] : 0.f;
                    dsc[r] += d[r] * x4[r];
                    dsh[r] += d[r];
                    d[r] *= sc4[r];
                }
                *(u32x2*)(a.dx + off) = u32x2{pack2(d[0], d[1]), pack2(d[2], d[3])};
            }
        }

        // ---- dT[j][k] = sum_i G[i][j] act(x)[i][k]:  D[m <-> k = 16 (m >> 2) + 4 kb + (m & 3)][n <-> j = 16 w + n],  contraction over rows
        f32x4 dtv[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) dtv[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x2 b0 = lds_tr_read(gimg + (32 * s + prow) * PT_PITCH + 16 * w + pcol);
            u32x2 b1 = lds_tr_read(gimg + (32 * s + 16 + prow) * PT_PITCH + 16 * w + pcol);
            u32x2 x0[4], x1[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {                      // the four 4-column pieces of the block sit 16 columns apart: lane m receives column 16 (m >> 2) + 4 kb + (m & 3)
                x0[kb] = lds_tr_read(ximg + (32 * s + prow) * PT_PITCH + 16 * (i16 & 3) + 4 * kb);
                x1[kb] = lds_tr_read(ximg + (32 * s + 16 + prow) * PT_PITCH + 16 * (i16 & 3) + 4 * kb);
            }
            lds_tr_wait(b0); lds_tr_wait(b1);
            const u32x4 bm = {b0.x, b0.y, b1.x, b1.y};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                lds_tr_wait(x0[kb]); lds_tr_wait(x1[kb]);
                const u32x4 am = {x0[kb].x, x0[kb].y, x1[kb].x, x1[kb].y};
                dtv[kb] = PPS_MFMA16(as_frag(am), as_frag(bm), dtv[kb], 0, 0, 0);
            }
        }
        {
            uint16_t* dst = a.dt + q * 4096 + (16 * w + i16) * 64 + 16 * kg;      // lane (j = 16 w + i16, kg): k = 16 kg + 4 kb + r
            *(u32x4*)dst = u32x4{pack2(dtv[0][0], dtv[0][1]), pack2(dtv[0][2], dtv[0][3]), pack2(dtv[1][0], dtv[1][1]), pack2(dtv[1][2], dtv[1][3])};
            *(u32x4*)(dst + 8) = u32x4{pack2(dtv[2][0], dtv[2][1]), pack2(dtv[2][2], dtv[2][3]), pack2(dtv[3][0], dtv[3][1]), pack2(dtv[3][2], dtv[3][3])};
        }
    }
    if (a.partials) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s0 = row16_sum(dsc[r]), s1 = row16_sum(dsh[r]);
            if (i16 == 0) {
                a.partials[(int64_t)blockIdx.x * 128 + 16 * w + 4 * kg + r] = s0;
                a.partials[(int64_t)blockIdx.x * 128 + 64 + 16 * w + 4 * kg + r] = s1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Input of the interpolation head (source/poco_model.py:400-404 with fc1 split into its latent and its offset part):
//     h1[(q, j), :] = table[ids[q, j], :] + Wx (query[q] - pts[ids[q, j]])          table [N, C] bf16, h1 [Q*k, C] bf16, Wx [C, 3]
// one 16-byte chunk (8 channels) per thread, written once (gather, offset, 3 -> C layer and the sum were four passes over [Q*k, C]).
// Backward: d table through the segmented sum of the gather (pps_segment_sum_rows_16), d Wx[c][d] = sum_rows dh1[row][c] * rel[row][d] here.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_input_fwd_kernel(const uint16_t* __restrict__ table, const int64_t* __restrict__ ids,
                                                            const float* __restrict__ pts, const float* __restrict__ query, int64_t rows, int k, int c,
                                                            const float* __restrict__ wx, uint16_t* __restrict__ h1) {
    const int cpr = c >> 3;                                   // chunks per row
    const int ch = threadIdx.x % cpr, rl = threadIdx.x / cpr, rpb = 256 / cpr;
    float w[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) { w[j][0] = wx[(8 * ch + j) * 3]; w[j][1] = wx[(8 * ch + j) * 3 + 1]; w[j][2] = wx[(8 * ch + j) * 3 + 2]; }
    for (int64_t row = (int64_t)blockIdx.x * rpb + rl; row < rows; row += (int64_t)gridDim.x * rpb) {
        const int64_t n = ids[row], q = row / k;
        const float r0 = query[q * 3] - pts[n * 3], r1 = query[q * 3 + 1] - pts[n * 3 + 1], r2 = query[q * 3 + 2] - pts[n * 3 + 2];
        const u32x4 t = *(const u32x4*)(table + n * c + 8 * ch);
        float e[8] = {lo16(t.x), hi16(t.x), lo16(t.y), hi16(t.y), lo16(t.z), hi16(t.z), lo16(t.w), hi16(t.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] += w[j][0] * r0 + w[j][1] * r1 + w[j][2] * r2;
        *(u32x4*)(h1 + row * c + 8 * ch) = u32x4{pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7])};
    }
}

// partial [block][c * 3]: d Wx
__global__ __launch_bounds__(256) void head_input_dwx_kernel(const uint16_t* __restrict__ dh1, const int64_t* __restrict__ ids,
                                                            const float* __restrict__ pts, const float* __restrict__ query, int64_t rows, int k, int c,
                                                            float* __restrict__ partials) {
    __shared__ float red[256][25];
    const int cpr = c >> 3;
    const int ch = threadIdx.x % cpr, rl = threadIdx.x / cpr, rpb = 256 / cpr;
    float acc[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j][0] = acc[j][1] = acc[j][2] = 0.f;
    constexpr int U = 4;                                       // four rows per thread in flight (the id, then the point it names: two dependent
                                                               // latencies per row; 0.20 -> 0.18 ms -- the forward kernel gains nothing from it); the sums keep their row order
    const int64_t stride = (int64_t)gridDim.x * rpb;
    for (int64_t row0 = (int64_t)blockIdx.x * rpb + rl; row0 < rows; row0 += U * stride) {
        int64_t rw[U], n[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rw[u] = row0 + u * stride;
            n[u] = ids[rw[u] < rows ? rw[u] : row0];
        }
        u32x4 t[U];
        float rel[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rr = rw[u] < rows ? rw[u] : row0, q = rr / k;
            t[u] = *(const u32x4*)(dh1 + rr * c + 8 * ch);
#pragma unroll
            for (int d = 0; d < 3; ++d) rel[u][d] = query[q * 3 + d] - pts[n[u] * 3 + d];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (rw[u] >= rows) break;
            const float g[8] = {lo16(t[u].x), hi16(t[u].x), lo16(t[u].y), hi16(t[u].y), lo16(t[u].z), hi16(t[u].z), lo16(t[u].w), hi16(t[u].w)};
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[j][0] += g[j] * rel[u][0]; acc[j][1] += g[j] * rel[u][1]; acc[j][2] += g[j] * rel[u][2]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x][3 * j] = acc[j][0]; red[threadIdx.x][3 * j + 1] = acc[j][1]; red[threadIdx.x][3 * j + 2] = acc[j][2]; }
    __syncthreads();
    for (int o = threadIdx.x; o < c * 3; o += 256) {             // output (channel, d): the threads of that channel's chunk in row-lane order
        const int cc = o / 3, d = o - 3 * cc;
        float t = 0.f;
        for (int r = 0; r < rpb; ++r) t += red[r * cpr + (cc >> 3)][3 * (cc & 7) + d];
        partials[(int64_t)blockIdx.x * c * 3 + o] = t;
    }
}

int g_cus = 0;
int cu_count() {
    if (g_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        g_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return g_cus;
}
int grid_for(int64_t units) {
    int g = cu_count();
    if (g > MAXP) g = MAXP;
    return (int)(units < g ? (units > 0 ? units : 1) : g);
}

template <typename K>
bool allow_lds(K kernel, size_t bytes) { return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess; }

template <int CK, int CO, bool DX, int POOL = 0>
int launch_layer(const LayerArgs& a, int grid, hipStream_t st) {
    static bool ok = allow_lds(rows_layer_kernel<CK, CO, DX, POOL>, layer_lds<CK, CO>());
    if (!ok) return PPS_ERR_LAUNCH;
    constexpr size_t lds = layer_lds<CK, CO>();
    hipLaunchKernelGGL((rows_layer_kernel<CK, CO, DX, POOL>), dim3(grid), dim3(NT), lds, st, a);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}
template <int CI, int CO, bool HAS_Y, int POOL = 0>
int launch_dw_y(const DwArgs& a, int grid, hipStream_t st) {
    static bool ok = allow_lds(rows_dw_kernel<CI, CO, HAS_Y, POOL>, dw_lds<CI, CO>());
    if (!ok) return PPS_ERR_LAUNCH;
    constexpr size_t lds = dw_lds<CI, CO>();
    hipLaunchKernelGGL((rows_dw_kernel<CI, CO, HAS_Y, POOL>), dim3(grid), dim3(NT), lds, st, a);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}
template <int CI, int CO>
int launch_dw(const DwArgs& a, int grid, hipStream_t st) {
    return a.y ? launch_dw_y<CI, CO, true>(a, grid, st) : launch_dw_y<CI, CO, false>(a, grid, st);
}

bool dim_ok(int c) { return c == 64 || c == 128 || c == 256; }

#define PPS_DISPATCH(CI, CO, CALL)                                            \
    do {                                                                      \
        const int key_ = (CI) * 1000 + (CO);                                  \
        switch (key_) {                                                       \
            case 64064: { constexpr int I = 64, O = 64; CALL; } break;        \
            case 64128: { constexpr int I = 64, O = 128; CALL; } break;       \
            case 64256: { constexpr int I = 64, O = 256; CALL; } break;       \
            case 128064: { constexpr int I = 128, O = 64; CALL; } break;      \
            case 128128: { constexpr int I = 128, O = 128; CALL; } break;     \
            case 128256: { constexpr int I = 128, O = 256; CALL; } break;     \
            case 256064: { constexpr int I = 256, O = 64; CALL; } break;      \
            case 256128: { constexpr int I = 256, O = 128; CALL; } break;     \
            case 256256: { constexpr int I = 256, O = 256; CALL; } break;     \
            default: return PPS_ERR_ARG;                                      \
        }                                                                     \
    } while (0)


// ---- per-type bodies of the C entries ------------------------------------------------------------------------------------------


int pps_rows_layer_supported(int cin, int cout) { return dim_ok(cin) && dim_ok(cout) ? 1 : 0; }

/* scratch of one call: slab partials of the statistics / input-affine sums, of dW and of db */
size_t pps_rows_layer_ws_bytes(int cin, int cout) {
    if (!dim_ok(cin) || !dim_ok(cout)) return 0;
    const size_t big = (size_t)(cin > cout ? cin : cout);
    return (size_t)MAXP * (2 * big + (size_t)cin * cout + cout) * sizeof(float) + (size_t)2 * cout * sizeof(float);
}

/* conv0a of PointNet in train(): y [rows, 64] bf16 = x [rows, 3] w^T + bias, batch statistics -> out_affine / save / running statistics as in
 * pps_rows_layer_fwd; backward: dw [64, 3], dbias [64] (NULL = skip), dgamma, dbeta (x gets no gradient: it is the input patch).
 * ws: pps_rows3_ws_bytes() bytes. */
size_t pps_rows3_ws_bytes() { return (size_t)((MAXP * 4 + 1) * 256 + 128 + 256) * sizeof(float); }

int pps_rows3_fwd(const float* x, int64_t rows, const float* w, const float* bias, void* y, const float* gamma, const float* beta,
                  float* running_mean, float* running_var, float momentum, float eps, float* out_affine, float* save, void* ws, void* stream) {
    if (rows < 1 || !x || !w || !y || !ws) return PPS_ERR_ARG;
    const bool bn = gamma != nullptr;
    if (bn && (!beta || !out_affine || !save)) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t blocks = (rows + R3_ROWS - 1) / R3_ROWS;
    const int grid = (int)(blocks < 4 * MAXP ? blocks : 4 * MAXP);
    hipLaunchKernelGGL(rows3_fwd_kernel, dim3(grid), dim3(256), 0, st, x, w, bias, rows, (uint16_t*)y, bn ? (float*)ws : nullptr);
    if (bn)
        hipLaunchKernelGGL(bn_affine_kernel, dim3(4), dim3(256), 0, st, (const float*)ws, grid, 64, (double)rows, gamma, beta, running_mean, running_var,
                           momentum, eps, out_affine, save);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_rows3_bwd(const float* x, const void* y, const void* gy, int64_t rows, const float* gamma, const float* save, const float* d_affine,
                  float* dw, float* dbias, float* dgamma, float* dbeta, void* ws, void* stream) {
    if (rows < 1 || !x || !gy || !dw || !ws) return PPS_ERR_ARG;
    const bool bn = gamma != nullptr;
    if (bn && (!y || !save || !d_affine || !dgamma || !dbeta)) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    float* total = part + (size_t)MAXP * 4 * 256;                 // [256]: dW [64][3] then db [64]
    float* gstat = total + 256;                                   // [2][64]
    if (bn) hipLaunchKernelGGL(bn_affine_bwd_kernel, dim3(1), dim3(64), 0, st, d_affine, save, gamma, 64, (double)rows, gstat, dgamma, dbeta);
    const int64_t blocks = (rows + R3_ROWS - 1) / R3_ROWS;
    const int grid = (int)(blocks < 4 * MAXP ? blocks : 4 * MAXP);
    hipLaunchKernelGGL(rows3_bwd_kernel, dim3(grid), dim3(256), 0, st, x, (const uint16_t*)y, (const uint16_t*)gy, bn ? gstat : nullptr, rows, part);
    launch_sum_partials(part, grid, 256, total, st);
    if (hipMemcpyAsync(dw, total, 192 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return PPS_ERR_LAUNCH;
    if (dbias && hipMemcpyAsync(dbias, total + 192, 64 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return PPS_ERR_LAUNCH;
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

/* Feature transform of PointNet in train() (source/base/nn.py:330-331): out[q,i,:] = act(x)[q,i,:] T[q]^T per group q of p <= 64 rows.
 * x [q*p, 64] bf16 stored activation with act = relu?(x * in_scale + in_shift) (NULL = identity), T [q, 64, 64] bf16 (+ I if add_identity),
 * out [q*p, 64] bf16.  Backward from g = d out: dx [q*p, 64] bf16, dt [q, 64, 64] bf16, d_in_affine [2][64] (NULL = skip).
 * ws: pps_patch_transform_ws_bytes() bytes. */
size_t pps_patch_transform_ws_bytes() { return (size_t)4 * MAXP * 128 * sizeof(float); }

int pps_patch_transform_fwd(const void* x, const float* in_scale, const float* in_shift, int in_relu, const void* t, int add_identity, int64_t q,
                            int p, void* out, void* stream) {
    if (q < 0 || p < 1 || p > 64 || ((in_scale == nullptr) != (in_shift == nullptr)) || (!in_scale && in_relu)) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!x || !t || !out) return PPS_ERR_ARG;
    PtArgs a{};
    a.x = (const uint16_t*)x; a.scale = in_scale; a.shift = in_shift; a.relu = in_relu; a.t = (const uint16_t*)t; a.add_identity = add_identity;
    a.nq = q; a.p = p; a.out = (uint16_t*)out;
    const int64_t blocks = (q + 3) / 4;
    const int grid = (int)(blocks < 8 * (int64_t)cu_count() ? blocks : 8 * (int64_t)cu_count());
    hipLaunchKernelGGL(patch_transform_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_patch_transform_bwd(const void* x, const float* in_scale, const float* in_shift, int in_relu, const void* t, int add_identity, const void* g,
                            int64_t q, int p, void* dx, void* dt, float* d_in_affine, void* ws, void* stream) {
    if (q < 0 || p < 1 || p > 64 || ((in_scale == nullptr) != (in_shift == nullptr)) || (!in_scale && in_relu)) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!x || !t || !g || !dx || !dt || !ws) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    PtArgs a{};
    a.x = (const uint16_t*)x; a.scale = in_scale; a.shift = in_shift; a.relu = in_relu; a.t = (const uint16_t*)t; a.add_identity = add_identity;
    a.nq = q; a.p = p; a.g = (const uint16_t*)g; a.dx = (uint16_t*)dx; a.dt = (uint16_t*)dt;
    a.partials = d_in_affine ? (float*)ws : nullptr;
    const int grid = (int)(q < 4 * (int64_t)MAXP ? q : 4 * (int64_t)MAXP);
    hipLaunchKernelGGL(patch_transform_bwd_kernel, dim3(grid), dim3(256), 0, st, a);
    if (d_in_affine) launch_sum_partials((const float*)ws, grid, 128, d_in_affine, st);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

/* Input of the interpolation head in train(): h1[(q,j),:] = table[ids[q,j],:] + wx (query[q] - pts[ids[q,j]]) (source/poco_model.py:400-404 with
 * fc1 split into latent and offset part).  table [n, c] bf16, ids [q*k] rows of table / pts, pts [n, 3], query [q, 3] fp32, wx [c, 3] fp32,
 * h1 [q*k, c] bf16; c a multiple of 8 dividing 2048.  dwx: d wx [c, 3] from dh1 [q*k, c] bf16; ws: pps_head_input_ws_bytes(c) bytes. */
size_t pps_head_input_ws_bytes(int c) { return c > 0 ? (size_t)4 * MAXP * c * 3 * sizeof(float) : 0; }

int pps_head_input_fwd(const void* table, const int64_t* ids, const float* pts, const float* query, int64_t q, int k, int c, const float* wx, void* h1,
                       void* stream) {
    if (q < 0 || k < 1 || c < 8 || (c & 7) || 256 % (c >> 3)) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!table || !ids || !pts || !query || !wx || !h1) return PPS_ERR_ARG;
    const int64_t rows = q * k, blocks = (rows + 256 / (c >> 3) - 1) / (256 / (c >> 3));
    const int grid = (int)(blocks < 16 * (int64_t)cu_count() ? blocks : 16 * (int64_t)cu_count());
    hipLaunchKernelGGL(head_input_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)table, ids, pts, query, rows, k, c, wx,
                       (uint16_t*)h1);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_head_input_dwx(const void* dh1, const int64_t* ids, const float* pts, const float* query, int64_t q, int k, int c, float* dwx, void* ws,
                       void* stream) {
    if (q < 1 || k < 1 || c < 8 || (c & 7) || 256 % (c >> 3)) return PPS_ERR_ARG;
    if (!dh1 || !ids || !pts || !query || !dwx || !ws) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = q * k, blocks = (rows + 256 / (c >> 3) - 1) / (256 / (c >> 3));
    const int grid = (int)(blocks < 4 * (int64_t)MAXP ? blocks : 4 * (int64_t)MAXP);
    hipLaunchKernelGGL(head_input_dwx_kernel, dim3(grid), dim3(256), 0, st, (const uint16_t*)dh1, ids, pts, query, rows, k, c, (float*)ws);
    launch_sum_partials((const float*)ws, grid, (int64_t)c * 3, dwx, st);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_rows_extrema_bf16(const void* x, int64_t groups, int p, int c, float* mx, float* mn, int* amx, int* amn, void* stream) {
    if (groups < 0 || p < 1 || c < 4 || (c & 3)) return PPS_ERR_ARG;
    if (groups == 0) return PPS_OK;
    if (!x || !mx || !mn || !amx || !amn) return PPS_ERR_ARG;
    const int64_t blocks = (groups + 3) / 4;
    const int grid = (int)(blocks < 8 * (int64_t)cu_count() ? blocks : 8 * (int64_t)cu_count());
    hipLaunchKernelGGL(rows_extrema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, groups, p, c, mx, mn, amx, amn);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_rows_layer_fwd(const void* x, int64_t rows, int cin, const float* in_scale, const float* in_shift, int in_relu, const float* w,
                       const float* bias, int cout, void* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float momentum, float eps, float* out_affine, float* save, void* ws, void* stream) {
    if (rows < 1 || !dim_ok(cin) || !dim_ok(cout)) return PPS_ERR_ARG;
    if (!x || !w || !y || !ws || ((in_scale == nullptr) != (in_shift == nullptr))) return PPS_ERR_ARG;
    const bool bn = gamma != nullptr;
    if (bn && (!beta || !out_affine || !save)) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for((rows + 31) / 32);
    LayerArgs a{};
    a.src = (const uint16_t*)x;
    a.pre_a = in_scale;
    a.pre_b = in_shift;
    a.pre_relu = in_relu;
    a.w = w;
    a.bias = bias;
    a.dst = (uint16_t*)y;
    a.partials = bn ? (float*)ws : nullptr;
    a.rows = rows;
    int rc = PPS_OK;
    PPS_DISPATCH(cin, cout, rc = (launch_layer<I, O, false>(a, grid, st)));
    if (rc != PPS_OK) return rc;
    if (bn) {
        hipLaunchKernelGGL(bn_affine_kernel, dim3((cout + 15) / 16), dim3(256), 0, st, (const float*)ws, grid, cout, (double)rows, gamma, beta,
                           running_mean, running_var, momentum, eps, out_affine, save);
        if (hipGetLastError() != hipSuccess) return PPS_ERR_LAUNCH;
    }
    return PPS_OK;
}

// the shapes whose gradient may arrive per group (pps_rows_layer_bwd_pooled): the last layer of PointNet's STN, 128 -> 256 with BatchNorm
bool pooled_ok(int cin, int cout, bool bn, int pool_p, int64_t rows) {
    return cin == 128 && cout == 256 && bn && pool_p >= 2 && pool_p <= 255 && rows % pool_p == 0 && rows * pool_p < (int64_t)1 << 32;
}

int rows_layer_bwd_any(const void* x, const void* y, const void* gy, const uint8_t* garg, int pool_p, int64_t rows, int cin, int cout,
                       const float* in_scale, const float* in_shift, int in_relu, const float* w, const float* gamma, const float* save,
                       const float* d_affine, void* dx, const void* dx_add, float* d_in_affine, float* dw, float* dbias, float* dgamma, float* dbeta,
                       void* ws, void* stream, const float* att_a = nullptr, const void* att_dp = nullptr, int att_k = 0,
                       const float* const* rank2 = nullptr /* {a, dl, dP, v}: the gradient rebuilt by att_grad8, gy and garg unused */) {
    if (rows < 1 || !dim_ok(cin) || !dim_ok(cout)) return PPS_ERR_ARG;
    if (att_a && (!att_dp || dx_add || in_scale || !in_relu || att_k < 1 || rows % att_k || rows * att_k >= (int64_t)1 << 32)) return PPS_ERR_ARG;
    if (!x || (!gy && !rank2) || !w || !ws || ((in_scale == nullptr) != (in_shift == nullptr))) return PPS_ERR_ARG;
    const bool bn = gamma != nullptr;
    if (bn && (!y || !save || !d_affine || !dgamma || !dbeta)) return PPS_ERR_ARG;
    const bool pool = pool_p != 0;
    if (pool && ((!garg && !rank2) || !pooled_ok(cin, cout, bn, pool_p, rows))) return PPS_ERR_ARG;
    if (rank2 && (!pool || !rank2[0] || !rank2[1] || !rank2[2] || !rank2[3])) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t big = (size_t)(cin > cout ? cin : cout);
    float* part_aff = (float*)ws;                                       // [MAXP][2][big]
    float* part_dw = part_aff + (size_t)MAXP * 2 * big;                 // [MAXP][cout][cin]
    float* part_db = part_dw + (size_t)MAXP * cin * cout;               // [MAXP][cout]
    float* gstat = part_db + (size_t)MAXP * cout;                       // [2][cout]
    if (bn) {
        hipLaunchKernelGGL(bn_affine_bwd_kernel, dim3((cout + 63) / 64), dim3(64), 0, st, d_affine, save, gamma, cout, (double)rows, gstat, dgamma,
                           dbeta);
        if (hipGetLastError() != hipSuccess) return PPS_ERR_LAUNCH;
    }
    int rc = PPS_OK;
    if (dx) {
        const int grid = grid_for((rows + 31) / 32);
        LayerArgs a{};
        a.src = (const uint16_t*)gy;
        a.src2 = bn ? (const uint16_t*)y : nullptr;
        a.pre_a = bn ? gstat : nullptr;
        a.pre_b = bn ? gstat + cout : nullptr;
        a.w = w;
        a.xin = (in_scale || in_relu) ? (const uint16_t*)x : nullptr;
        a.in_scale = in_scale;
        a.in_shift = in_shift;
        a.in_relu = in_relu;
        a.dst = (uint16_t*)dx;
        a.addend = (const uint16_t*)dx_add;
        a.att_a = att_a;
        a.att_dp = (const uint16_t*)att_dp;
        a.att_k = att_k;
        a.att_magic = att_a ? (unsigned)(0x100000000ull / (unsigned)att_k) + 1u : 0u;
        a.partials = d_in_affine ? part_aff : nullptr;
        a.rows = rows;
        a.garg = garg;
        a.pool_p = pool_p;
        a.pool_magic = pool ? (unsigned)(0x100000000ull / (unsigned)pool_p) + 1u : 0u;
        if (rank2) { a.g_a = rank2[0]; a.g_dl = rank2[1]; a.g_dp = rank2[2]; a.g_v = rank2[3]; }
        if (rank2) rc = launch_layer<256, 128, true, 2>(a, grid, st);
        else if (pool) rc = launch_layer<256, 128, true, 1>(a, grid, st);
        else PPS_DISPATCH(cin, cout, rc = (launch_layer<O, I, true>(a, grid, st)));
        if (rc != PPS_OK) return rc;
        if (d_in_affine) {
            launch_sum_partials(part_aff, grid, (int64_t)2 * cin, d_in_affine, st);
            if (hipGetLastError() != hipSuccess) return PPS_ERR_LAUNCH;
        }
    }
    if (dw) {
        int kr = 32;
        PPS_DISPATCH(cin, cout, kr = 32 * (dw_ksteps<I, O>()));
        const int64_t nsteps = (rows + kr - 1) / kr;
        int grid = grid_for(nsteps);
        const int64_t per = (nsteps + grid - 1) / grid;
        grid = (int)((nsteps + per - 1) / per);                          // no empty slabs
        DwArgs a{};
        a.gy = (const uint16_t*)gy;
        a.y = bn ? (const uint16_t*)y : nullptr;
        a.gs = bn ? gstat : nullptr;
        a.gq2 = bn ? gstat + cout : nullptr;
        a.x = (const uint16_t*)x;
        a.in_scale = in_scale;
        a.in_shift = in_shift;
        a.in_relu = in_relu;
        a.dw_part = part_dw;
        a.db_part = dbias ? part_db : nullptr;
        a.rows = rows;
        a.garg = garg;
        a.pool_p = pool_p;
        a.pool_magic = pool ? (unsigned)(0x100000000ull / (unsigned)pool_p) + 1u : 0u;
        if (rank2) { a.g_a = rank2[0]; a.g_dl = rank2[1]; a.g_dp = rank2[2]; a.g_v = rank2[3]; }
        if (rank2) rc = launch_dw_y<128, 256, true, 2>(a, grid, st);
        else if (pool) rc = launch_dw_y<128, 256, true, 1>(a, grid, st);
        else PPS_DISPATCH(cin, cout, rc = (launch_dw<I, O>(a, grid, st)));
        if (rc != PPS_OK) return rc;
        const int64_t nw = (int64_t)cin * cout;
        launch_sum_partials(part_dw, grid, nw, dw, st);
        if (dbias) launch_sum_partials(part_db, grid, (int64_t)cout, dbias, st);
        if (hipGetLastError() != hipSuccess) return PPS_ERR_LAUNCH;
    }
    return PPS_OK;
}

int pps_rows_layer_bwd(const void* x, const void* y, const void* gy, int64_t rows, int cin, int cout, const float* in_scale,
                       const float* in_shift, int in_relu, const float* w, const float* gamma, const float* save, const float* d_affine,
                       void* dx, const void* dx_add, float* d_in_affine, float* dw, float* dbias, float* dgamma, float* dbeta, void* ws,
                       void* stream) {
    return rows_layer_bwd_any(x, y, gy, nullptr, 0, rows, cin, cout, in_scale, in_shift, in_relu, w, gamma, save, d_affine, dx, dx_add, d_in_affine, dw,
                              dbias, dgamma, dbeta, ws, stream);
}

int pps_rows_layer_bwd_pooled(const void* x, const void* y, const void* gval, const uint8_t* garg, int pool_p, int64_t rows, int cin, int cout,
                              const float* in_scale, const float* in_shift, int in_relu, const float* w, const float* gamma, const float* save,
                              const float* d_affine, void* dx, float* d_in_affine, float* dw, float* dbias, float* dgamma, float* dbeta, void* ws,
                              void* stream) {
    if (pool_p == 0) return PPS_ERR_ARG;
    return rows_layer_bwd_any(x, y, gval, garg, pool_p, rows, cin, cout, in_scale, in_shift, in_relu, w, gamma, save, d_affine, dx, nullptr, d_in_affine,
                              dw, dbias, dgamma, dbeta, ws, stream);
}

int pps_rows_layer_bwd_attn(const void* x, const void* gy, int64_t rows, int cin, int cout, const float* w, const float* att_weights, const void* att_dpooled,
                            int att_k, void* dx, float* dw, float* dbias, void* ws, void* stream) {
    if (!att_weights || !dx) return PPS_ERR_ARG;
    return rows_layer_bwd_any(x, nullptr, gy, nullptr, 0, rows, cin, cout, nullptr, nullptr, 1, w, nullptr, nullptr, nullptr, dx, nullptr, nullptr, dw, dbias,
                              nullptr, nullptr, ws, stream, att_weights, att_dpooled, att_k);
}

int pps_rows_layer_bwd_rank2(const void* x, const void* y, const float* g_a, const float* g_dl, const float* g_dp, const float* g_v, int pool_p, int64_t rows,
                             int cin, int cout, const float* in_scale, const float* in_shift, int in_relu, const float* w, const float* gamma,
                             const float* save, const float* d_affine, void* dx, float* d_in_affine, float* dw, float* dbias, float* dgamma, float* dbeta,
                             void* ws, void* stream) {
    if (pool_p == 0) return PPS_ERR_ARG;
    const float* const rank2[4] = {g_a, g_dl, g_dp, g_v};
    return rows_layer_bwd_any(x, y, nullptr, nullptr, pool_p, rows, cin, cout, in_scale, in_shift, in_relu, w, gamma, save, d_affine, dx, nullptr, d_in_affine,
                              dw, dbias, dgamma, dbeta, ws, stream, nullptr, nullptr, 0, rank2);
}

int pps_rows_layer_pooled_supported(int cin, int cout, int pool_p) { return pooled_ok(cin, cout, true, pool_p, (int64_t)pool_p) ? 1 : 0; }


#include "pps_head_chain_impl.h"

#undef PPS_MFMA16
#undef PPS_DISPATCH
}  // namespace PPS_NS
