// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of ppsurf_amd.
//
// Register-resident activation tiles for the f32-input MFMA `v_mfma_f32_16x16x4_f32`:
// one wave owns a tile of 16 ROWS (neighbours / patch points / queries) x C channels and keeps it in
// registers in the MFMA C/D layout, so that the output of one dense layer IS the B operand of the next
// layer without any data movement (the K index of the contraction is permuted consistently in the
// packed weights).  Block b (16 channels) is one f32x4 per lane:
//     lane l = (n = l & 15, g = l >> 4), register r   <->   row n, channel 16*b + 4*g + r.
// Weights are packed on the host (pps_pack.cpp) as  [ob][kb][lane][4]  floats with
//     packed[ob][kb][l][s] = W[16*ob + (l & 15)][16*kb + 4*(l >> 4) + s]
// so one ds_read_b128 / global_load_dwordx4 per lane feeds the A operand of 4 consecutive MFMAs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PPS_WAVE 64
#define PPS_OK 0
#define PPS_ERR_ARG 1
#define PPS_ERR_LAUNCH 2

// scheduling groups (llvm.amdgcn.sched.group.barrier masks)
#define PPS_SG_MFMA 0x008
#define PPS_SG_DSREAD 0x100
#define PPS_SG_VALU 0x002

namespace pps {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---- reductions over the 16 rows of a tile (lanes with equal l>>4), pure DPP ------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// after these 4 steps every lane of a 16-lane row holds the reduction over the row
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_mov<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_mov<0x141>(v));   // row_half_mirror
    v = fmaxf(v, dpp_mov<0x140>(v));   // row_mirror
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}

// max(a, b) as ONE v_max_f32.  fmaxf of two values the compiler cannot prove canonical (loaded, or out of inline assembly) is three instructions
// in IEEE mode -- a quieting v_max x, x per operand in front of the maximum -- and __builtin_amdgcn_fmed3f(a, b, inf) is folded back into the same
// maxnum: 912 of the 1360 v_max_f32 of pointnet_stn_rows_kernel<true> were such quieting moves (its running maximum over the patch rows).
// NaNs: v_max_f32 returns the other operand for a quiet NaN like fmaxf does.
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The same reductions for FOUR values at once with the lane permutation folded INTO the arithmetic instruction (v_max_f32_dpp / v_add_f32_dpp: the
// first source is read through the DPP permutation).  Written with dpp_mov the compiler keeps a separate v_mov_b32_dpp per step -- it pairs the
// adds of two values into v_pk_add_f32, which has no DPP form, and puts a canonicalising v_max in front of every fmaxf -- 2.5-3 instructions
// per step and value instead of 1 (profiles/round6_interp_isa_census.md).  Inline assembly is outside the compiler's hazard recogniser: a DPP
// read of a VGPR needs 2 wait states after the VALU write of it.  Inside a block the four independent chains are interleaved (a dependent
// instruction is 4 issues behind its producer) and the block starts with s_nop 1 for whatever the compiler scheduled right in front of it.
#define PPS_DPP4(OP, CTRL)                                                    \
    OP " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      \
    OP " %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      \
    OP " %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      \
    OP " %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void row16_max4(float& a, float& b, float& c, float& d) {
    asm("s_nop 1\n\t" PPS_DPP4("v_max_f32_dpp", "quad_perm:[1,0,3,2]") PPS_DPP4("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
        PPS_DPP4("v_max_f32_dpp", "row_half_mirror") PPS_DPP4("v_max_f32_dpp", "row_mirror")
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void row16_sum4(float& a, float& b, float& c, float& d) {
    asm("s_nop 1\n\t" PPS_DPP4("v_add_f32_dpp", "quad_perm:[1,0,3,2]") PPS_DPP4("v_add_f32_dpp", "quad_perm:[2,3,0,1]")
        PPS_DPP4("v_add_f32_dpp", "row_half_mirror") PPS_DPP4("v_add_f32_dpp", "row_mirror")
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
#undef PPS_DPP4

// ---- value of lane (l ^ st) without the LDS crossbar -------------------------------------------------
// A bitonic network over the 64 lanes is a chain of dependent exchanges; as ds_bpermute each costs an LDS round trip (the kNN kernels spend
// more time in their merge networks than in the distance tests).  Strides 1 and 2 are quad permutations, 4 and 8 two DPP moves
// (l ^ 4 = (l ^ 7) ^ 3: row_half_mirror, then the quad reversed;  l ^ 8 = (l ^ 15) ^ 7: row_mirror, then row_half_mirror), 16 and 32 the
// gfx950 row / half swaps (v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second,
// v_permlane32_swap the upper half of the first with the lower half of the second: with the same value in both, the partner's value is in
// the second result for lanes whose stride bit is clear and in the first for the others).  `st` must be a compile-time constant after
// unrolling for the switch to fold.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned lane_xor_u32(unsigned v, int st, int lane) {
    switch (st) {
    case 1: return dpp_mov_u32<0xB1>(v);
    case 2: return dpp_mov_u32<0x4E>(v);
    case 4: return dpp_mov_u32<0x1B>(dpp_mov_u32<0x141>(v));
    case 8: return dpp_mov_u32<0x141>(dpp_mov_u32<0x140>(v));
    case 16: { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (lane & 16) ? r[0] : r[1]; }
    case 32: { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane & 32) ? r[0] : r[1]; }
    default: return (unsigned)__shfl_xor((int)v, st);
    }
}
__device__ __forceinline__ unsigned long long lane_xor_u64(unsigned long long v, int st, int lane) {
    return ((unsigned long long)lane_xor_u32((unsigned)(v >> 32), st, lane) << 32) | lane_xor_u32((unsigned)v, st, lane);
}

// Sum over the 16 rows of EVERY channel of a 16-block tile (64 registers) in 240 instead of 576 VALU ops: a butterfly in
// which each step halves the number of live registers (lanes with the step's bit set keep the upper half of the blocks and
// hand the lower half to their partner, and vice versa).  On return lane (n,g) holds in v[0] the sums of block n:
// channels 16*n + 4*g + r, r = 0..3 -> one contiguous ds_write_b128 / global store per lane.
template <int CTRL>
__device__ __forceinline__ f32x4 dpp_mov4(f32x4 v) { return f32x4{dpp_mov<CTRL>(v.x), dpp_mov<CTRL>(v.y), dpp_mov<CTRL>(v.z), dpp_mov<CTRL>(v.w)}; }

template <int HALF, int CTRL>
__device__ __forceinline__ void butterfly_step(f32x4* v, bool upper) {
#pragma unroll
    for (int b = 0; b < HALF; ++b) {
        const f32x4 keep = upper ? v[b + HALF] : v[b];
        const f32x4 send = upper ? v[b] : v[b + HALF];
        v[b] = keep + dpp_mov4<CTRL>(send);
    }
}
__device__ __forceinline__ void rows16_sum_transposed(f32x4 (&v)[16], int lane) {
    butterfly_step<8, 0x128>(v, (lane & 8) != 0);    // row_ror:8   partner n ^ 8
    butterfly_step<4, 0x141>(v, (lane & 4) != 0);    // row_half_mirror   partner flips bits 0..2
    butterfly_step<2, 0x4E>(v, (lane & 2) != 0);     // quad_perm [2,3,0,1]   partner n ^ 2
    butterfly_step<1, 0xB1>(v, (lane & 1) != 0);     // quad_perm [1,0,3,2]   partner n ^ 1
}

// the same for an 8-block tile (128 channels): lanes n and n ^ 8 are added first, then three halving steps; lane (n,g) ends with block n & 7
__device__ __forceinline__ void rows16_sum_transposed8(f32x4 (&v)[8], int lane) {
#pragma unroll
    for (int b = 0; b < 8; ++b) v[b] += dpp_mov4<0x128>(v[b]);   // row_ror:8
    butterfly_step<4, 0x141>(v, (lane & 4) != 0);
    butterfly_step<2, 0x4E>(v, (lane & 2) != 0);
    butterfly_step<1, 0xB1>(v, (lane & 1) != 0);
}

// ---- dense layer on a register tile ------------------------------------------------------------------
// out[ob] = act( bias + sum_kb W[ob][kb] * in[kb] ),  weights read through `w` (LDS or global, packed layout,
// pointing at the first f32x4 of (ob = OB0, kb = 0) for this lane's chunk), two output blocks in flight.
// ACT: 0 none, 1 relu.  INIT: 0 accumulator starts from the bias, 1 from the current contents of out[].
template <int KB, int NOB, int ACT, int INIT = 0>
__device__ __forceinline__ void dense_blocks(const f32x4 (&in)[KB], f32x4* out, const f32x4* __restrict__ w,
                                             const f32x4* __restrict__ bias, int lane) {
    static_assert(NOB % 2 == 0, "output blocks are processed in pairs");
    const int g = lane >> 4;
#pragma unroll
    for (int ob = 0; ob < NOB; ob += 2) {
        f32x4 acc0 = INIT ? out[ob] : bias[(ob) * 4 + g];
        f32x4 acc1 = INIT ? out[ob + 1] : bias[(ob + 1) * 4 + g];
        // A fragments are requested TWO k-steps ahead: the pair consumed by step kb was issued during step kb-2, so the
        // wait in front of its first MFMA is a counted lgkmcnt with the newer pair still in flight (no exposed LDS latency)
        f32x4 p0 = w[((ob) * KB) * 64 + lane];
        f32x4 p1 = w[((ob + 1) * KB) * 64 + lane];
        f32x4 q0 = p0, q1 = p1;
        if (KB > 1) {
            q0 = w[((ob) * KB + 1) * 64 + lane];
            q1 = w[((ob + 1) * KB + 1) * 64 + lane];
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const f32x4 a0 = p0, a1 = p1;
            p0 = q0; p1 = q1;
            if (kb + 2 < KB) {
                q0 = w[((ob) * KB + kb + 2) * 64 + lane];
                q1 = w[((ob + 1) * KB + kb + 2) * 64 + lane];
            }
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, in[kb].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, in[kb].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, in[kb].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, in[kb].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, in[kb].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, in[kb].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, in[kb].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, in[kb].w, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(PPS_SG_DSREAD, 2, 0);
            __builtin_amdgcn_sched_group_barrier(PPS_SG_MFMA, 8, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ACT == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0[r] = fmaxf(acc0[r], 0.f);
                acc1[r] = fmaxf(acc1[r], 0.f);
            }
        }
        out[ob] = acc0;
        out[ob + 1] = acc1;
    }
}

// ---- split-precision dense layer: fp32-grade products on the f16 matrix pipe (opt-in decoder dtype "f16x3") ----------
// x = hi + lo with hi = f16(x) (round toward zero, v_cvt_pkrtz_f16_f32) and lo = f16(x - hi) (the subtraction is exact), so
// hi + lo carries ~21 significant bits; W = Whi + Wlo likewise (round to nearest, packed on the host).  One product needs three
// MFMAs, Whi.hi + Whi.lo + Wlo.hi (the lo.lo term is below 2^-21), each v_mfma_f32_16x16x32_f16 doing 8x the work of
// v_mfma_f32_16x16x4_f32 in half its issue time: 16/3 of the fp32 matrix rate.  Measured error of the whole decoder on the
// reference's golden logits: 1e-6 at logit scale 1, 1.2e-5 at scale 28 (tools/split_precision_study.py) -- the level of the fp32
// path's own rounding noise.
// Layout: the C/D tile of the 16x16 MFMA is the same as for the f32 instruction (lane (n,g), register r <-> row n, channel
// 16 ob + 4 g + r), so two finished output blocks (2 kb, 2 kb + 1) ARE the B operand of k-block kb of the next layer once
// split: element j of lane (n,g) is channel 32 kb + 16 (j >> 2) + 4 g + (j & 3).  The packed weights use the same map:
//     packed[ob][kb][part][lane][j] = f16 part (0: hi, 1: lo) of W[16 ob + (l & 15)][32 kb + 16 (j >> 2) + 4 (l >> 4) + (j & 3)].
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct HiLo { half8 hi, lo; };

// The low part comes from ONE mixed-precision fma per value: v_fma_mixlo_f16 / v_fma_mixhi_f16 read the f16 half of `hi` selected by op_sel as
// its first operand, compute x - hi in fp32 (exact: hi is x truncated to 11 significant bits) and write the result, converted to f16 (round to
// nearest even), into the low / high half of the destination -- 1.5 VALU instructions per value (half a v_cvt_pkrtz + one fma_mix) instead of 3
// (v_cvt_pkrtz, v_cvt_f32_f16 back, v_sub_f32, v_cvt_pkrtz of the difference: rounds 2-5).  The split is 25 % of all vector instructions of
// interp_pool_f16x3_kernel (profiles/round6_interp_isa_census.md).  As inline assembly: hipcc does not form the mix instructions from the
// fptrunc(fsub(x, fpext(h))) pattern on its own.
__device__ __forceinline__ HiLo split_f16(const f32x4& x0, const f32x4& x1) {
    u32x4 hp, lp;
    const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[2 * p], v[2 * p + 1]));
        unsigned l;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(v[2 * p]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(v[2 * p + 1]));
        hp[p] = h;
        lp[p] = l;
    }
    return HiLo{__builtin_bit_cast(half8, hp), __builtin_bit_cast(half8, lp)};
}
// Range guard of the split: hi = f16(x) needs |x| <= 65504 (v_cvt_pkrtz saturates silently beyond it, no inf appears anywhere).  Every kernel that splits
// activations keeps the running maximum magnitude of what it split (4 v_max3_f32 per pair of blocks) and raises a flag in device memory when the
// range was left; the host-side entry point (pps_decode_fwd_mixed_f32) queues the exact-fp32 kernels behind the split-precision ones, gated on
// that flag, so a chunk that left the range is recomputed in fp32 without a host round trip.
#define PPS_F16_MAX 65504.f
__device__ __forceinline__ void range_track(float& amax, const f32x4& a, const f32x4& b) {
    // as inline assembly on purpose: written with fmaxf the compiler turns the running maximum into a reduction tree evaluated at the END of the
    // unrolled layer and keeps every fp32 output block alive until then (164 spilled VGPRs in interp_pool_f16x3_kernel)
    asm volatile("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|" : "+v"(amax) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));
    asm volatile("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|" : "+v"(amax) : "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
}
__device__ __forceinline__ HiLo split_f16_r(float& amax, const f32x4& x0, const f32x4& x1) {
    range_track(amax, x0, x1);
    return split_f16(x0, x1);
}
__device__ __forceinline__ void range_commit(float amax, int* flag) {
    if (flag != nullptr && !(amax <= PPS_F16_MAX)) atomicOr(flag, 1);
}
// gate of the fp32 fallback kernels: nothing to do unless a split-precision kernel of this chunk left the f16 range
__device__ __forceinline__ bool gate_closed(const int* gate) { return gate != nullptr && __builtin_nontemporal_load(gate) == 0; }

// back to fp32 (hi + lo), blocks 2 kb and 2 kb + 1: one v_fma_mix_f32 per value (f16 half of hi x 1.0 + f16 half of lo, in fp32) instead of two
// conversions and an add
__device__ __forceinline__ void join_f16(const HiLo& x, f32x4& y0, f32x4& y1) {
    const u32x4 hp = __builtin_bit_cast(u32x4, x.hi), lp = __builtin_bit_cast(u32x4, x.lo);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float a, b;
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(a) : "v"(hp[p]), "v"(lp[p]));
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(b) : "v"(hp[p]), "v"(lp[p]));
        if (p < 2) { y0[2 * p] = a; y0[2 * p + 1] = b; } else { y1[2 * (p - 2)] = a; y1[2 * (p - 2) + 1] = b; }
    }
}

// out blocks OB0 .. OB0+NOB-1 (NOB even) of act(bias + W in): in[kb] = split activations of k-block kb (KB blocks of 32 channels),
// w -> packed half8 fragments of (ob = OB0, kb = 0) for this wave: [ob][kb][hi|lo][64 lanes].  The result of each output-block
// PAIR is handed to `sink(pair_index, f32x4 block_even, f32x4 block_odd)`.
// Main products (hi.hi) and corrections (hi.lo + lo.hi) accumulate separately: dependent MFMAs are 4 issues apart, and the small
// terms are summed among themselves before they meet the large ones.
// FENCE = false drops the scheduling fence after each k-step: for short contractions (KB = 2) the epilogue of one pair (split +
// stores) then overlaps the MFMAs of the next pair instead of running alone.
// INIT = true: the main accumulators start from init[ob] (fp32 blocks of an earlier partial product over other input channels) instead of
// the bias -- a layer whose input is the concatenation of two tensors is evaluated half by half.
// `hook(ob, kb)` is called after the six MFMAs of every k-step have been issued: the place to issue ONE piece of the next weight chunk's copy
// (see stream_step_spread in pps_decode.hip) -- the wave's own MFMAs are queued in the matrix pipe while the memory pipeline accepts the piece.
// NP: f16 products per fp32 product -- 3 (hi.hi + hi.lo + lo.hi, the product's arithmetic), 2 (the weight's low part dropped) or 1 (plain f16):
// the reduced forms exist for ONE measured experiment on fc_query (tools/fcq_products.md); nothing in the product instantiates them.
template <int KB, int NOB, int ACT, bool FENCE = true, bool INIT = false, int NP = 3, class Sink, class Hook>
__device__ __forceinline__ void dense_blocks_f16x3_hook(const HiLo (&in)[KB], const half8* __restrict__ w, const f32x4* __restrict__ bias, int lane,
                                                        Sink&& sink, Hook&& hook, const f32x4* init = nullptr) {
    static_assert(NOB % 2 == 0, "output blocks are processed in pairs");
    const int g = lane >> 4;
#pragma unroll
    for (int ob = 0; ob < NOB; ob += 2) {
        f32x4 m0 = INIT ? init[ob] : bias[(ob) * 4 + g], m1 = INIT ? init[ob + 1] : bias[(ob + 1) * 4 + g];
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        const half8* w0 = w + ((ob) * KB) * 128 + lane;
        const half8* w1 = w + ((ob + 1) * KB) * 128 + lane;
        half8 ph0 = w0[0], pl0 = w0[64], ph1 = w1[0], pl1 = w1[64];          // fragments of step kb are requested during step kb-1
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const half8 ah0 = ph0, al0 = pl0, ah1 = ph1, al1 = pl1;
            if (kb + 1 < KB) {
                ph0 = w0[(kb + 1) * 128]; pl0 = w0[(kb + 1) * 128 + 64];
                ph1 = w1[(kb + 1) * 128]; pl1 = w1[(kb + 1) * 128 + 64];
            }
            m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, in[kb].hi, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, in[kb].hi, m1, 0, 0, 0);
            if (NP >= 2) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, in[kb].lo, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, in[kb].lo, c1, 0, 0, 0);
            }
            if (NP >= 3) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, in[kb].hi, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, in[kb].hi, c1, 0, 0, 0);
            }
            hook(ob, kb);
            if (FENCE) {
                __builtin_amdgcn_sched_group_barrier(PPS_SG_DSREAD, NP >= 3 ? 4 : 2, 0);
                __builtin_amdgcn_sched_group_barrier(PPS_SG_MFMA, 2 * NP, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        f32x4 o0 = m0 + c0, o1 = m1 + c1;
        if (ACT == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { o0[r] = fmaxf(o0[r], 0.f); o1[r] = fmaxf(o1[r], 0.f); }
        }
        sink(ob >> 1, o0, o1);
    }
}
template <int KB, int NOB, int ACT, bool FENCE = true, bool INIT = false, class Sink>
__device__ __forceinline__ void dense_blocks_f16x3(const HiLo (&in)[KB], const half8* __restrict__ w, const f32x4* __restrict__ bias, int lane,
                                                   Sink&& sink, const f32x4* init = nullptr) {
    dense_blocks_f16x3_hook<KB, NOB, ACT, FENCE, INIT>(in, w, bias, lane, sink, [](int, int) {}, init);
}

// The same layer for T row tiles of ONE wave at once (T = 2: 32 rows): every A fragment read from LDS feeds 3 T MFMAs instead of 3, and a streamed
// weight chunk (with its barrier) is paid once per T tiles.  Possible where the per-tile state is small -- the PointNet row kernels: 64 / 128-channel
// activations, 16 / 32 VGPRs per tile -- not in the 256-channel interpolation layers.  sink(tile, pair, o0, o1).
template <int T, int KB, int NOB, int ACT, bool FENCE = true, class Sink>
__device__ __forceinline__ void dense_blocks_f16x3_tiles(const HiLo (&in)[T][KB], const half8* __restrict__ w, const f32x4* __restrict__ bias, int lane,
                                                         Sink&& sink) {
    static_assert(NOB % 2 == 0, "output blocks are processed in pairs");
    const int g = lane >> 4;
#pragma unroll
    for (int ob = 0; ob < NOB; ob += 2) {
        f32x4 m0[T], m1[T], c0[T], c1[T];
        const f32x4 b0 = bias[(ob) * 4 + g], b1 = bias[(ob + 1) * 4 + g];
#pragma unroll
        for (int t = 0; t < T; ++t) { m0[t] = b0; m1[t] = b1; c0[t] = f32x4{0.f, 0.f, 0.f, 0.f}; c1[t] = c0[t]; }
        const half8* w0 = w + ((ob) * KB) * 128 + lane;
        const half8* w1 = w + ((ob + 1) * KB) * 128 + lane;
        half8 ph0 = w0[0], pl0 = w0[64], ph1 = w1[0], pl1 = w1[64];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const half8 ah0 = ph0, al0 = pl0, ah1 = ph1, al1 = pl1;
            if (kb + 1 < KB) {
                ph0 = w0[(kb + 1) * 128]; pl0 = w0[(kb + 1) * 128 + 64];
                ph1 = w1[(kb + 1) * 128]; pl1 = w1[(kb + 1) * 128 + 64];
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                m0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, in[t][kb].hi, m0[t], 0, 0, 0);
                m1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, in[t][kb].hi, m1[t], 0, 0, 0);
                c0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, in[t][kb].lo, c0[t], 0, 0, 0);
                c1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, in[t][kb].lo, c1[t], 0, 0, 0);
                c0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, in[t][kb].hi, c0[t], 0, 0, 0);
                c1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, in[t][kb].hi, c1[t], 0, 0, 0);
            }
            if (FENCE) {
                __builtin_amdgcn_sched_group_barrier(PPS_SG_DSREAD, 4, 0);
                __builtin_amdgcn_sched_group_barrier(PPS_SG_MFMA, 6 * T, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            f32x4 o0 = m0[t] + c0[t], o1 = m1[t] + c1[t];
            if (ACT == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { o0[r] = fmaxf(o0[r], 0.f); o1[r] = fmaxf(o1[r], 0.f); }
            }
            sink(t, ob >> 1, o0, o1);
        }
    }
}

// first layer for xyz inputs (K = 3 padded to 4): B operand of lane (n,g) is coordinate g of row n (0 for g = 3).
// wxyz packed [ob][lane]:  W[16*ob + (l & 15)][l >> 4]  (0 for l >> 4 == 3).
template <int NOB>
__device__ __forceinline__ void xyz_blocks(float coord, f32x4* acc, const float* __restrict__ wxyz, int lane) {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxyz[ob * 64 + lane], coord, acc[ob], 0, 0, 0);
}

// ---- cooperative weight-chunk streaming global -> LDS (LDS-DMA, no VGPR staging) ----------------------
// A chunk is NF4 * NTHREADS f32x4 (NF4 * NTHREADS * 16 bytes), copied verbatim with global_load_lds_dwordx4:
// each wave instruction moves 1 KiB to a wave-uniform LDS base + lane * 16.  The copy is asynchronous; the
// __syncthreads() that ends a pipeline step waits for it (vmcnt(0)) and publishes it to the other waves.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// The copy instruction is written as inline assembly on purpose.  hipcc models `global_load_lds` (a FLAT-encoded instruction that touches both
// the vector-memory and the LDS counters) as a "pending flat" access: while one is in flight EVERY later `s_waitcnt` on vmcnt or lgkmcnt is
// emitted with count 0.  The weight stream is always in flight, so every A-fragment wait of the MFMA loops became `s_waitcnt lgkmcnt(0)` -- it
// also drained the fragments just requested for the NEXT k-step, i.e. the LDS latency was exposed once per k-step instead of hidden two steps
// ahead (105 of 116 waits in interp_pool_f16x3_kernel; with the stream compiled out the same source gets lgkmcnt(4..7)).  That, not the stream's
// bandwidth, held the f16x3 kernels at ~50 % and the fp32 kernels at ~83 % of the matrix pipe (round 3).  As inline asm the
// compiler does not see a FLAT LDS access; the counters stay in order for its own bookkeeping: its vmcnt waits for ordinary loads only ever wait
// LONGER with unknown older/younger VMEM operations in flight (returns are in order), and the stream itself is waited for explicitly
// (stream_wait(), before the barrier that publishes a chunk).
template <int NF4, int NTHREADS>
__device__ __forceinline__ void chunk_copy_async(const f32x4* __restrict__ src, f32x4* dst) {
    // `src` is workgroup-uniform (SGPR base); the only per-lane part is one 32-bit byte offset
    unsigned lane_off = threadIdx.x * 16u;
    asm volatile("" : "+v"(lane_off));     // opaque: keeps hipcc from hoisting (and then spilling) one 64-bit address per chunk
    const int wave_base = threadIdx.x & ~63;
    const char* sbase = (const char*)src;
    asm volatile("" : "+s"(sbase));        // opaque as well: otherwise one SGPR pair per piece of every chunk is hoisted out of the persistent loop
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
        f32x4* d = dst + i * NTHREADS + wave_base;
        // wave-uniform LDS byte address -> M0 (the low half of a generic pointer into the LDS aperture is the LDS offset)
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)d);
        const char* piece = sbase + (size_t)(i * NTHREADS * 16);
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_addr), "v"(lane_off), "s"(piece) : "memory", "m0");
    }
}
// one 1 KiB-per-wave piece (index i of NF4) of the same copy: lets a caller spread the pieces of a chunk over its compute phase
template <int NTHREADS>
__device__ __forceinline__ void chunk_copy_piece(const f32x4* __restrict__ src, f32x4* dst, int i) {
    unsigned lane_off = threadIdx.x * 16u;
    asm volatile("" : "+v"(lane_off));
    const int wave_base = threadIdx.x & ~63;
    const char* piece = (const char*)src + (size_t)i * (NTHREADS * 16);
    asm volatile("" : "+s"(piece));
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dst + i * NTHREADS + wave_base));
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_addr), "v"(lane_off), "s"(piece) : "memory", "m0");
}
// The same piece with NO per-piece vector instruction: `voff` = this lane's byte offset INCLUDING the piece's (threadIdx.x * 16 + i * NTHREADS * 16),
// one VGPR per piece index made by stream_lane_offsets() once per kernel, and `chunk` = the chunk's first byte (one SGPR pair for all pieces of a
// chunk, advanced by the caller).  chunk_copy_piece builds its lane offset and its source pointer per piece: the opaque "+v" copy of the offset is
// a v_mov per piece, and the 18 x 4 piece pointers of a pass over the weights, all loop invariant, were hoisted out of the persistent loop and
// spilled -- v_writelane / v_readlane: 227 of the 2264 vector instructions of a trip of interp_pool_f16x3_kernel
// (profiles/round6_interp_isa_census.md).
template <int NTHREADS, int NPIECES>
__device__ __forceinline__ void stream_lane_offsets(unsigned (&voff)[NPIECES]) {
#pragma unroll
    for (int i = 0; i < NPIECES; ++i) {
        voff[i] = threadIdx.x * 16u + (unsigned)i * (NTHREADS * 16u);
        asm volatile("" : "+v"(voff[i]));          // opaque: stays one live VGPR instead of being rebuilt (or folded into 64-bit addresses) per use
    }
}
template <int NTHREADS>
__device__ __forceinline__ void chunk_copy_piece_at(const char* chunk, f32x4* dst, int i, unsigned voff) {
    const int wave_base = threadIdx.x & ~63;
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dst + i * NTHREADS + wave_base));
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_addr), "v"(voff), "s"(chunk) : "memory", "m0");
}
// all outstanding pieces of this wave have landed in LDS (call before the barrier that hands the chunk to the other waves)
__device__ __forceinline__ void stream_wait() {
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
}

// XCD-aware persistent tile order: workgroup b runs on XCD b % 8 (observed, speed only); each XCD walks a
// contiguous range of tiles so neighbouring queries share its L2.
__device__ __forceinline__ void xcd_tile_range(int ntiles, int& first, int& count, int& stride) {
    const int nwg = gridDim.x;
    const int b = blockIdx.x;
    if ((nwg & 7) != 0 || nwg < 8) { first = b; stride = nwg; count = (ntiles > b) ? (ntiles - b + nwg - 1) / nwg : 0; return; }
    const int xcd = b & 7, slot = b >> 3, per = nwg >> 3;
    const int lo = (int)(((int64_t)ntiles * xcd) / 8), hi = (int)(((int64_t)ntiles * (xcd + 1)) / 8);
    first = lo + slot; stride = per;
    count = (hi - lo > slot) ? (hi - lo - slot + per - 1) / per : 0;
}

}  // namespace pps
