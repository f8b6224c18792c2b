// Exact brute-force k-nearest-neighbour search and patch normalisation for gfx950.
//
// replaces the CPU kd-tree of the reference: source/poco_utils.py:257-273 `knn`,
// source/base/proximity.py:40-89, source/poco_utils.py:67-72, source/ppsurf_data_loader.py:91-123.
//
// Design (HBM/L2-bound integer+fp32 VALU work, no MFMA):
//   * one wave scans the whole cloud for QW queries at a time: each lane loads ONE point per step (coalesced,
//     the 1.2 MB cloud stays in the XCD's L2) and evaluates it against QW wave-uniform queries;
//   * every query keeps its current k best as a SORTED list spread over the 64 lanes (lane i = i-th best) of a
//     64-bit key (d2 bits << 32 | index): unsigned key order == (d2, index) lexicographic order;
//   * a point is a candidate only if d2 < tau (the current k-th d2).  Scanning in ascending index order makes the
//     strict test exact for ties.  Candidates are compacted into a small LDS buffer with ballot/mbcnt and merged
//     into the list by a wave-wide bitonic sort + merge when the buffer fills.
//   * d2 = ((dx*dx + dy*dy) + dz*dz) with explicit round-to-nearest mul/add (no FMA contraction) so indices are
//     bit-identical to the CPU oracle.
#include "pps_common.h"
#include "../../include/ppsurf_amd.h"

#define KNN_QW 8            // queries per wave pass
#define KNN_WAVES 4         // waves per workgroup
#define KNN_CAP 128         // candidate buffer entries per query (flush threshold 64 + one step of 64)
#ifndef KNN_BISECT
#define KNN_BISECT 12        // halvings of the distance-bit range when the seed bound is bisected
#endif
#ifndef KNN_SEED
#define KNN_SEED 4           // full blocks around the best window whose exact k-th distance seeds tau (knn_blocked_kernel; at least R are taken)
#endif

typedef unsigned long long u64;

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
    const int lo = __shfl_xor((int)(unsigned)v, m), hi = __shfl_xor((int)(unsigned)(v >> 32), m);
    return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
    const int lo = __shfl((int)(unsigned)v, src), hi = __shfl((int)(unsigned)(v >> 32), src);
    return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}

// minimum over the 64 lanes, in every lane: DPP row reduction + the row / half swaps of gfx950 (pps_common.h)
__device__ __forceinline__ float wave_min_f32(float v, int lane) {
    v = fminf(v, pps::dpp_mov<0xB1>(v));
    v = fminf(v, pps::dpp_mov<0x4E>(v));
    v = fminf(v, pps::dpp_mov<0x141>(v));
    v = fminf(v, pps::dpp_mov<0x140>(v));
    v = fminf(v, __uint_as_float(pps::lane_xor_u32(__float_as_uint(v), 16, lane)));
    v = fminf(v, __uint_as_float(pps::lane_xor_u32(__float_as_uint(v), 32, lane)));
    return v;
}

// ascending bitonic sort of one key per lane over the 64 lanes
__device__ __forceinline__ u64 wave_sort64(u64 key, int lane) {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int st = size >> 1; st > 0; st >>= 1) {
            const u64 other = pps::lane_xor_u64(key, st, lane);
            const bool up = ((lane & size) == 0);          // ascending block
            const bool lower = ((lane & st) == 0);
            const bool take_min = (up == lower);
            const u64 mn = key < other ? key : other, mx = key < other ? other : key;
            key = take_min ? mn : mx;
        }
    }
    return key;
}
// `key` is bitonic over the 64 lanes -> ascending
__device__ __forceinline__ u64 wave_bitonic_merge64(u64 key, int lane) {
#pragma unroll
    for (int st = 32; st > 0; st >>= 1) {
        const u64 other = pps::lane_xor_u64(key, st, lane);
        const bool lower = ((lane & st) == 0);
        const u64 mn = key < other ? key : other, mx = key < other ? other : key;
        key = lower ? mn : mx;
    }
    return key;
}

// merge `cnt` buffered candidates into the sorted list; returns the new list
__device__ __forceinline__ u64 knn_flush(u64 list, const u64* cand, int cnt, int lane) {
    while (cnt > 0) {
        const int c = cnt < 64 ? cnt : 64;
        u64 ck = (lane < c) ? cand[cnt - c + lane] : ~0ull;
        ck = wave_sort64(ck, lane);
        const u64 rev = shfl_u64(ck, 63 - lane);
        list = wave_bitonic_merge64(list < rev ? list : rev, lane);
        cnt -= c;
    }
    return list;
}

__global__ __launch_bounds__(KNN_WAVES * 64) void knn_kernel(const float* __restrict__ pts, int n, const float* __restrict__ query,
                                                             int64_t m, int k, int64_t* __restrict__ out_idx,
                                                             float* __restrict__ out_d2) {
    __shared__ u64 cand_all[KNN_WAVES][KNN_QW][KNN_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ntask = (m + KNN_QW - 1) / KNN_QW;
    for (int64_t task = (int64_t)blockIdx.x * KNN_WAVES + wave; task < ntask; task += (int64_t)gridDim.x * KNN_WAVES) {
        const int64_t q0 = task * KNN_QW;
        float qx[KNN_QW], qy[KNN_QW], qz[KNN_QW], tau[KNN_QW];
        u64 list[KNN_QW];
        int cnt[KNN_QW];
#pragma unroll
        for (int j = 0; j < KNN_QW; ++j) {
            const int64_t qq = (q0 + j < m) ? q0 + j : m - 1;
            qx[j] = __builtin_nontemporal_load(query + qq * 3);
            qy[j] = query[qq * 3 + 1];
            qz[j] = query[qq * 3 + 2];
            qx[j] = __shfl(qx[j], 0); qy[j] = __shfl(qy[j], 0); qz[j] = __shfl(qz[j], 0);
            tau[j] = INFINITY;
            list[j] = ~0ull;
            cnt[j] = 0;
        }
        for (int base = 0; base < n; base += 64) {
            const int p = base + lane;
            const bool pv = p < n;
            const int pc = pv ? p : n - 1;
            const float px = pts[3 * pc], py = pts[3 * pc + 1], pz = pts[3 * pc + 2];
#pragma unroll
            for (int j = 0; j < KNN_QW; ++j) {
                const float dx = __fsub_rn(qx[j], px), dy = __fsub_rn(qy[j], py), dz = __fsub_rn(qz[j], pz);
                const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                const bool pass = pv && (d2 < tau[j]);
                const u64 mask = __ballot(pass);
                if (mask != 0ull) {
                    u64* cand = cand_all[wave][j];
                    const int pos = cnt[j] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                    if (pass) cand[pos] = ((u64)__float_as_uint(d2) << 32) | (unsigned)p;
                    cnt[j] += __popcll(mask);
                    if (cnt[j] > KNN_CAP - 64) {
                        list[j] = knn_flush(list[j], cand + (cnt[j] - 64), 64, lane);      // the newest 64 only: one sorting network per flush.  tau then
                        cnt[j] -= 64;                                                     // comes from a subset of the points seen: looser, still exact
                        tau[j] = __uint_as_float((unsigned)(shfl_u64(list[j], k - 1) >> 32));
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < KNN_QW; ++j) {
            list[j] = knn_flush(list[j], cand_all[wave][j], cnt[j], lane);
            if (q0 + j < m && lane < k) {
                out_idx[(q0 + j) * k + lane] = (int64_t)(unsigned)(list[j] & 0xffffffffull);
                if (out_d2) out_d2[(q0 + j) * k + lane] = __uint_as_float((unsigned)(list[j] >> 32));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Blocked variant: the same exhaustive search over a cloud that was reordered along a space-filling curve and cut into
// blocks of 64 points with axis-aligned bounding boxes.  A block is skipped for a query when its box lower bound
// near2 = ((gx*gx + gy*gy) + gz*gz)  (g = per-axis gap, evaluated with the SAME rounding order as d2, hence never above
// the d2 of a point inside the box: rounding is monotonic) exceeds the current k-th distance tau.  tau starts from a box
// UPPER bound: the farthest-corner distance of any full block bounds the k-th distance (k <= 64 points lie inside).
// Results are bit-identical to knn_kernel: candidates are tested with d2 <= tau and ordered by the full (d2, original
// index) key in the merge network.
// ---------------------------------------------------------------------------------------------------------------
// two sorted 64-lists -> lower (returned in a) and upper (returned in b) sorted halves of their union
__device__ __forceinline__ void wave_merge2(u64& a, u64& b, int lane) {
    const u64 rev = shfl_u64(b, 63 - lane);
    const u64 lo = a < rev ? a : rev, hi = a < rev ? rev : a;
    a = wave_bitonic_merge64(lo, lane);
    b = wave_bitonic_merge64(hi, lane);
}

// sorted list of 64*R keys, element e = r*64 + lane; merge the buffered candidates, keep the 64*R smallest
template <int R>
__device__ __forceinline__ void knn_flush_r(u64 (&list)[R], const u64* cand, int cnt, int lane) {
    while (cnt > 0) {
        const int c = cnt < 64 ? cnt : 64;
        u64 carry = (lane < c) ? cand[cnt - c + lane] : ~0ull;
        carry = wave_sort64(carry, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) wave_merge2(list[r], carry, lane);
        cnt -= c;
    }
}

template <int R>
__global__ __launch_bounds__(KNN_WAVES * 64) void knn_blocked_kernel(const float* __restrict__ pts, const int* __restrict__ orig,
                                                                     const float* __restrict__ bbox, int nb,
                                                                     const float* __restrict__ win_bbox, int n_win,
                                                                     const float* __restrict__ sbox /* boxes of 64 consecutive blocks, or null */,
                                                                     const float* __restrict__ query, int64_t m, int k,
                                                                     int64_t* __restrict__ out_idx, float* __restrict__ out_d2) {
    __shared__ u64 cand_all[KNN_WAVES][KNN_QW][KNN_CAP];
    const int ngrp = (nb + 63) >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kr = (k - 1) >> 6, kl = (k - 1) & 63;          // register / lane of the k-th key
    const int64_t ntask = (m + KNN_QW - 1) / KNN_QW;
    for (int64_t task = (int64_t)blockIdx.x * KNN_WAVES + wave; task < ntask; task += (int64_t)gridDim.x * KNN_WAVES) {
        const int64_t q0 = task * KNN_QW;
        float qx[KNN_QW], qy[KNN_QW], qz[KNN_QW], tau[KNN_QW];
        u64 list[KNN_QW][R];
        int cnt[KNN_QW], wbest[KNN_QW];
#pragma unroll
        for (int j = 0; j < KNN_QW; ++j) {
            const int64_t qq = (q0 + j < m) ? q0 + j : m - 1;
            qx[j] = __shfl(query[qq * 3], 0); qy[j] = __shfl(query[qq * 3 + 1], 0); qz[j] = __shfl(query[qq * 3 + 2], 0);
            tau[j] = INFINITY;
            wbest[j] = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) list[j][r] = ~0ull;
            cnt[j] = 0;
        }
        // ---- initial tau: the farthest-corner distance of a window of R full blocks (>= k points) bounds the k-th distance -----------------
        // Any window gives a valid bound; a good one is wanted.  With the boxes of 64-block groups at hand (`sbox`) only the windows of ONE group
        // per query are looked at -- the group whose own farthest corner is nearest -- instead of all of them (1563 boxes for a 100k-point cloud,
        // a third of this kernel's instructions); without, every window is, as in rounds 1-3.
        if (sbox != nullptr && n_win > 0) {
#pragma unroll
            for (int j = 0; j < KNN_QW; ++j) {
                float best = INFINITY;
                int bw = 0;
                for (int g0 = 0; g0 * 64 < n_win; g0 += 64) {                    // groups that hold at least one window
                    const int g = g0 + lane;
                    const bool gv = g * 64 < n_win;
                    const float* gb = sbox + (int64_t)(gv ? g : 0) * 6;
                    const float gx = fmaxf(fmaxf(__fsub_rn(gb[0], qx[j]), __fsub_rn(qx[j], gb[3])), 0.f);
                    const float gy = fmaxf(fmaxf(__fsub_rn(gb[1], qy[j]), __fsub_rn(qy[j], gb[4])), 0.f);
                    const float gz = fmaxf(fmaxf(__fsub_rn(gb[2], qz[j]), __fsub_rn(qz[j], gb[5])), 0.f);
                    const float gnear = gv ? __fadd_rn(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)), __fmul_rn(gz, gz)) : INFINITY;
                    // a window's farthest corner is not nearer than its group's box: only groups nearer than the best bound so far can improve it
                    u64 pend = __ballot(gnear < best);
                    while (pend != 0ull) {
                        const bool mine = (pend >> lane) & 1ull;
                        const float tn = wave_min_f32(mine ? gnear : INFINITY, lane);
                        const int src = __builtin_ctzll(__ballot(mine && gnear == tn));          // the nearest pending group first
                        const int b = (g0 + src) * 64 + lane;
                        const bool bv = b < n_win;
                        const float* bb = win_bbox + (int64_t)(bv ? b : 0) * 6;
                        const float fx = fmaxf(fabsf(__fsub_rn(qx[j], bb[0])), fabsf(__fsub_rn(qx[j], bb[3])));
                        const float fy = fmaxf(fabsf(__fsub_rn(qy[j], bb[1])), fabsf(__fsub_rn(qy[j], bb[4])));
                        const float fz = fmaxf(fabsf(__fsub_rn(qz[j], bb[2])), fabsf(__fsub_rn(qz[j], bb[5])));
                        const float far2 = bv ? __fadd_rn(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)), __fmul_rn(fz, fz)) : INFINITY;
                        const float tf = wave_min_f32(far2, lane);
                        if (tf < best) {
                            best = tf;
                            bw = (g0 + src) * 64 + __builtin_ctzll(__ballot(far2 == tf));
                        }
                        pend &= ~(1ull << src);
                        pend &= __ballot(gnear < best);
                    }
                }
                tau[j] = best;                                                   // wave-uniform: the reduction below leaves both as they are
                wbest[j] = bw;
            }
        } else {
            for (int b0 = 0; b0 < n_win; b0 += 64) {
                const int b = b0 + lane;
                const bool bv = b < n_win;
                const float* bb = win_bbox + (int64_t)(bv ? b : 0) * 6;
                const float lx = bb[0], ly = bb[1], lz = bb[2], hx = bb[3], hy = bb[4], hz = bb[5];
#pragma unroll
                for (int j = 0; j < KNN_QW; ++j) {
                    const float fx = fmaxf(fabsf(__fsub_rn(qx[j], lx)), fabsf(__fsub_rn(qx[j], hx)));
                    const float fy = fmaxf(fabsf(__fsub_rn(qy[j], ly)), fabsf(__fsub_rn(qy[j], hy)));
                    const float fz = fmaxf(fabsf(__fsub_rn(qz[j], lz)), fabsf(__fsub_rn(qz[j], hz)));
                    const float far2 = bv ? __fadd_rn(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)), __fmul_rn(fz, fz)) : INFINITY;
                    if (far2 < tau[j]) { tau[j] = far2; wbest[j] = b; }   // per lane: the minimum over its windows (and which one) ...
                }
            }
        }
        // ... and ONE reduction over the lanes per query (it used to sit inside the window loop: 48 ds_bpermute per 64 windows)
        // ---- seed: the farthest-corner bound of a window is about the distance of its LAST point, loose by the whole extent of a 64-point cell;
        // with it the scan below buffered and sorted several hundred candidates per query before tau closed in.  The exact k-th smallest distance
        // among the 64 * nseed points of the full blocks around the best window is a far tighter upper bound of the k-th distance and costs no sort:
        // a 31-step bisection on the distance bits with wave ballots (distances are non-negative floats: their bit patterns order like the values).
        const int nfull = n_win > 0 ? n_win + R - 1 : 0;                          // windows slide over the FULL blocks (ops.KnnBlocks._windows)
        constexpr int SEEDS = KNN_SEED > R ? KNN_SEED : R;                        // 64 * SEEDS >= k
        const int nseed = nfull < SEEDS ? nfull : SEEDS;                          // >= R whenever there is a window
#pragma unroll
        for (int j = 0; j < KNN_QW; ++j) {
            const float t = wave_min_f32(tau[j], lane);
            const u64 who = __ballot(tau[j] == t);
            tau[j] = t;
            if (nseed > 0) {
                const int w = __shfl(wbest[j], who != 0ull ? __builtin_ctzll(who) : 0);
                int s0 = w - (nseed - R) / 2;
                s0 = s0 < 0 ? 0 : s0;
                s0 = s0 + nseed > nfull ? nfull - nseed : s0;
                unsigned u[SEEDS];
#pragma unroll
                for (int rr = 0; rr < SEEDS; ++rr) {
                    const int p = (s0 + (rr < nseed ? rr : 0)) * 64 + lane;
                    const float dx = __fsub_rn(qx[j], pts[3 * p]), dy = __fsub_rn(qy[j], pts[3 * p + 1]), dz = __fsub_rn(qz[j], pts[3 * p + 2]);
                    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                    u[rr] = rr < nseed ? __float_as_uint(d2) : 0xffffffffu;
                }
                // invariant: #{u <= hi} >= k (the window bound to start with: its >= k points are among the seeds or it is no better than them);
                // KNN_BISECT halvings of the bit range leave hi within ~2^-6 of the exact k-th distance, and any hi is a valid bound
                unsigned lo = 0u, hi = __float_as_uint(tau[j]);
                {
                    int c = 0;
#pragma unroll
                    for (int rr = 0; rr < SEEDS; ++rr) c += __popcll(__ballot(u[rr] <= hi));
                    if (c < k) lo = hi;                                           // the seeds do not hold k points below the window bound: keep it
                }
#pragma unroll
                for (int it = 0; it < KNN_BISECT; ++it) {
                    const unsigned mid = lo + ((hi - lo) >> 1);
                    int c = 0;
#pragma unroll
                    for (int rr = 0; rr < SEEDS; ++rr) c += __popcll(__ballot(u[rr] <= mid));
                    if (c >= k) hi = mid; else lo = mid + (lo < hi ? 1u : 0u);
                }
                tau[j] = __uint_as_float(hi);
            }
        }
        // ---- scan: groups of 64 blocks whose box a query's ball reaches (all groups without `sbox`), 64 block boxes per culling step inside a
        // group, surviving blocks point by point.  The group test is the block test on a larger box: same rounding order, never above the lower
        // bound of any block inside (the gaps to the union box are no larger, every operation is monotonic).
        for (int g0 = 0; g0 < ngrp; g0 += 64) {
          u64 gneed[KNN_QW];
          u64 gany = 0ull;
          {
            const int g = g0 + lane;
            const bool gv = g < ngrp;
            if (sbox != nullptr) {
                const float* bb = sbox + (int64_t)(gv ? g : 0) * 6;
                const float lx = bb[0], ly = bb[1], lz = bb[2], hx = bb[3], hy = bb[4], hz = bb[5];
#pragma unroll
                for (int j = 0; j < KNN_QW; ++j) {
                    const float gx = fmaxf(fmaxf(__fsub_rn(lx, qx[j]), __fsub_rn(qx[j], hx)), 0.f);
                    const float gy = fmaxf(fmaxf(__fsub_rn(ly, qy[j]), __fsub_rn(qy[j], hy)), 0.f);
                    const float gz = fmaxf(fmaxf(__fsub_rn(lz, qz[j]), __fsub_rn(qz[j], hz)), 0.f);
                    const float near2 = __fadd_rn(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)), __fmul_rn(gz, gz));
                    gneed[j] = __ballot(gv && near2 <= tau[j]);
                    gany |= gneed[j];
                }
            } else {
                gany = __ballot(gv);
#pragma unroll
                for (int j = 0; j < KNN_QW; ++j) gneed[j] = gany;
            }
          }
          while (gany != 0ull) {
            const int gbit = __builtin_ctzll(gany);
            gany &= gany - 1ull;
            const int b0 = (g0 + gbit) * 64;
            const int b = b0 + lane;
            const bool bv = b < nb;
            const float* bb = bbox + (int64_t)(bv ? b : 0) * 6;
            const float lx = bb[0], ly = bb[1], lz = bb[2], hx = bb[3], hy = bb[4], hz = bb[5];
            u64 need[KNN_QW];
            u64 any = 0ull;
#pragma unroll
            for (int j = 0; j < KNN_QW; ++j) {
                need[j] = 0ull;
                if (((gneed[j] >> gbit) & 1ull) == 0ull) continue;
                const float gx = fmaxf(fmaxf(__fsub_rn(lx, qx[j]), __fsub_rn(qx[j], hx)), 0.f);
                const float gy = fmaxf(fmaxf(__fsub_rn(ly, qy[j]), __fsub_rn(qy[j], hy)), 0.f);
                const float gz = fmaxf(fmaxf(__fsub_rn(lz, qz[j]), __fsub_rn(qz[j], hz)), 0.f);
                const float near2 = __fadd_rn(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)), __fmul_rn(gz, gz));
                need[j] = __ballot(bv && near2 <= tau[j]);
                any |= need[j];
            }
            while (any != 0ull) {
                const int bit = __builtin_ctzll(any);
                any &= any - 1ull;
                const int p = (b0 + bit) * 64 + lane;
                const float px = pts[3 * p], py = pts[3 * p + 1], pz = pts[3 * p + 2];
                const int oi = orig[p];
                const bool pv = oi >= 0;
#pragma unroll
                for (int j = 0; j < KNN_QW; ++j) {
                    if (((need[j] >> bit) & 1ull) == 0ull) continue;
                    const float dx = __fsub_rn(qx[j], px), dy = __fsub_rn(qy[j], py), dz = __fsub_rn(qz[j], pz);
                    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                    const bool pass = pv && (d2 <= tau[j]);
                    const u64 mask = __ballot(pass);
                    if (mask != 0ull) {
                        u64* cand = cand_all[wave][j];
                        const int pos = cnt[j] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                        if (pass) cand[pos] = ((u64)__float_as_uint(d2) << 32) | (unsigned)oi;
                        cnt[j] += __popcll(mask);
                        if (cnt[j] > KNN_CAP - 64) {
                            knn_flush_r<R>(list[j], cand + (cnt[j] - 64), 64, lane);      // the newest 64: one full sorting network; the rest waits
                            cnt[j] -= 64;
                            // the register that holds the k-th key, picked by compile-time comparisons: indexing `list[j]` with the run-time kr put
                            // the whole [KNN_QW][R] array into scratch memory for R >= 3 (208 / 272 bytes per lane; every merge network of the
                            // k = 200 search of config 5 then went through memory: 415 MB written per 25 000-query launch for 40 MB of results,
                            // profiles/round6_config5_pmc.json at commit 1b7b9f8)
                            u64 kth = list[j][R - 1];
#pragma unroll
                            for (int r = 0; r + 1 < R; ++r) kth = (r == kr) ? list[j][r] : kth;
                            tau[j] = fminf(tau[j], __uint_as_float((unsigned)(shfl_u64(kth, kl) >> 32)));
                        }
                    }
                }
            }
          }
        }
#pragma unroll
        for (int j = 0; j < KNN_QW; ++j) {
            knn_flush_r<R>(list[j], cand_all[wave][j], cnt[j], lane);
            if (q0 + j < m) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e = r * 64 + lane;
                    if (e < k) {
                        out_idx[(q0 + j) * k + e] = (int64_t)(unsigned)(list[j][r] & 0xffffffffull);
                        if (out_d2) out_d2[(q0 + j) * k + e] = __uint_as_float((unsigned)(list[j][r] >> 32));
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Batched blocked search (k <= 64): the large neighbourhood tables of a whole batch of clouds in ONE launch.  A "kind" is one table
// of the encoder (e.g. ids01: points of level 0, queries of level 1) for all B clouds of the batch; every cloud's level has been
// arranged by the caller like for knn_blocked_kernel (Morton order, blocks of 64, boxes), all clouds of a kind with the same sizes and
// a constant stride between them.  Queries may be given in Morton order too (q_orig maps the position to the output row): the 8
// queries of a wave are then neighbours in space and share the blocks they have to visit.
// replaces: the 13 kd-tree builds + queries per cloud of source/poco_data_loader.py:155-168 for the tables whose point set is large.
// ---------------------------------------------------------------------------------------------------------------
#define KBB_MAX 8
struct KbbKind {
    const float* pts; const int* orig; const float* bbox; const float* win; const float* query; const int* q_orig; int64_t* out;
    int64_t pts_stride, orig_stride, bbox_stride, win_stride, query_stride, qorig_stride, out_stride;      // elements between clouds
    int nb, n_win, m, k, groups, group_end;
};
struct KbbArgs { KbbKind kind[KBB_MAX]; int nkinds, nclouds; };

__global__ __launch_bounds__(KNN_WAVES * 64) void knn_blocked_batch_kernel(const KbbArgs a) {
    __shared__ u64 cand_all[KNN_WAVES][KNN_QW][KNN_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = a.kind[a.nkinds - 1].group_end;
    for (int grp = blockIdx.x * KNN_WAVES + wave; grp < total; grp += gridDim.x * KNN_WAVES) {
        int t = 0;
        while (grp >= a.kind[t].group_end) ++t;
        const KbbKind& kd = a.kind[t];
        const int local = grp - (t == 0 ? 0 : a.kind[t - 1].group_end);
        const int cloud = local / kd.groups, q0 = (local % kd.groups) * KNN_QW;
        const float* __restrict__ pts = kd.pts + cloud * kd.pts_stride;
        const int* __restrict__ orig = kd.orig + cloud * kd.orig_stride;
        const float* __restrict__ bbox = kd.bbox + cloud * kd.bbox_stride;
        const float* __restrict__ win_bbox = kd.win + cloud * kd.win_stride;
        const float* __restrict__ query = kd.query + cloud * kd.query_stride;
        const int nb = kd.nb, n_win = kd.n_win, m = kd.m, k = kd.k;
        float qx[KNN_QW], qy[KNN_QW], qz[KNN_QW], tau[KNN_QW];
        u64 list[KNN_QW];
        int cnt[KNN_QW];
#pragma unroll
        for (int j = 0; j < KNN_QW; ++j) {
            const int qq = (q0 + j < m) ? q0 + j : m - 1;
            qx[j] = __shfl(query[qq * 3], 0); qy[j] = __shfl(query[qq * 3 + 1], 0); qz[j] = __shfl(query[qq * 3 + 2], 0);
            tau[j] = INFINITY;
            list[j] = ~0ull;
            cnt[j] = 0;
        }
        for (int b0 = 0; b0 < n_win; b0 += 64) {                  // initial tau: farthest corner of any FULL block (>= k points inside)
            const int b = b0 + lane;
            const bool bv = b < n_win;
            const float* bb = win_bbox + (int64_t)(bv ? b : 0) * 6;
            const float lx = bb[0], ly = bb[1], lz = bb[2], hx = bb[3], hy = bb[4], hz = bb[5];
#pragma unroll
            for (int j = 0; j < KNN_QW; ++j) {
                const float fx = fmaxf(fabsf(__fsub_rn(qx[j], lx)), fabsf(__fsub_rn(qx[j], hx)));
                const float fy = fmaxf(fabsf(__fsub_rn(qy[j], ly)), fabsf(__fsub_rn(qy[j], hy)));
                const float fz = fmaxf(fabsf(__fsub_rn(qz[j], lz)), fabsf(__fsub_rn(qz[j], hz)));
                const float far2 = bv ? __fadd_rn(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)), __fmul_rn(fz, fz)) : INFINITY;
                tau[j] = fminf(tau[j], far2);                      // per lane: the minimum over its windows ...
            }
        }
#pragma unroll
        for (int j = 0; j < KNN_QW; ++j) tau[j] = wave_min_f32(tau[j], lane);      // ... and ONE reduction over the lanes per query (it used to sit
                                                                                   // inside the window loop: 48 ds_bpermute per 64 windows)
        for (int b0 = 0; b0 < nb; b0 += 64) {
            const int b = b0 + lane;
            const bool bv = b < nb;
            const float* bb = bbox + (int64_t)(bv ? b : 0) * 6;
            const float lx = bb[0], ly = bb[1], lz = bb[2], hx = bb[3], hy = bb[4], hz = bb[5];
            u64 need[KNN_QW];
            u64 any = 0ull;
#pragma unroll
            for (int j = 0; j < KNN_QW; ++j) {
                const float gx = fmaxf(fmaxf(__fsub_rn(lx, qx[j]), __fsub_rn(qx[j], hx)), 0.f);
                const float gy = fmaxf(fmaxf(__fsub_rn(ly, qy[j]), __fsub_rn(qy[j], hy)), 0.f);
                const float gz = fmaxf(fmaxf(__fsub_rn(lz, qz[j]), __fsub_rn(qz[j], hz)), 0.f);
                const float near2 = __fadd_rn(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)), __fmul_rn(gz, gz));
                need[j] = __ballot(bv && near2 <= tau[j]);
                any |= need[j];
            }
            while (any != 0ull) {
                const int bit = __builtin_ctzll(any);
                any &= any - 1ull;
                const int p = (b0 + bit) * 64 + lane;
                const float px = pts[3 * p], py = pts[3 * p + 1], pz = pts[3 * p + 2];
                const int oi = orig[p];
                const bool pv = oi >= 0;
#pragma unroll
                for (int j = 0; j < KNN_QW; ++j) {
                    if (((need[j] >> bit) & 1ull) == 0ull) continue;
                    const float dx = __fsub_rn(qx[j], px), dy = __fsub_rn(qy[j], py), dz = __fsub_rn(qz[j], pz);
                    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                    const bool pass = pv && (d2 <= tau[j]);
                    const u64 mask = __ballot(pass);
                    if (mask != 0ull) {
                        u64* cand = cand_all[wave][j];
                        const int pos = cnt[j] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                        if (pass) cand[pos] = ((u64)__float_as_uint(d2) << 32) | (unsigned)oi;
                        cnt[j] += __popcll(mask);
                        if (cnt[j] > KNN_CAP - 64) {
                            list[j] = knn_flush(list[j], cand + (cnt[j] - 64), 64, lane);      // the newest 64 only (one sorting network)
                            cnt[j] -= 64;
                            tau[j] = fminf(tau[j], __uint_as_float((unsigned)(shfl_u64(list[j], k - 1) >> 32)));
                        }
                    }
                }
            }
        }
        const int* __restrict__ q_orig = kd.q_orig ? kd.q_orig + cloud * kd.qorig_stride : nullptr;
        int64_t* __restrict__ out = kd.out + cloud * kd.out_stride;
#pragma unroll
        for (int j = 0; j < KNN_QW; ++j) {
            list[j] = knn_flush(list[j], cand_all[wave][j], cnt[j], lane);
            if (q0 + j < m && lane < k) {
                const int row = q_orig ? q_orig[q0 + j] : q0 + j;
                out[(int64_t)row * k + lane] = (int64_t)(unsigned)(list[j] & 0xffffffffull);
            }
        }
    }
}

// one wave per query: lane j < P handles neighbour j
__global__ __launch_bounds__(256) void patch_normalize_kernel(const float* __restrict__ raw, const float* __restrict__ query,
                                                              const int64_t* __restrict__ idx, int64_t idx_stride, int64_t Q, int P,
                                                              float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= Q) return;
    const float qx = query[qi * 3], qy = query[qi * 3 + 1], qz = query[qi * 3 + 2];
    float mx = 0.f;
    for (int j = lane; j < P; j += 64) {
        const int64_t i = idx[qi * idx_stride + j];
        const float dx = __fsub_rn(raw[i * 3], qx), dy = __fsub_rn(raw[i * 3 + 1], qy), dz = __fsub_rn(raw[i * 3 + 2], qz);
        mx = fmaxf(mx, __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s));
    const float radius = __fsqrt_rn(mx);       // max_j ||p_j - q||  (ppsurf_data_loader.py:107-109)
    for (int j = lane; j < P; j += 64) {
        const int64_t i = idx[qi * idx_stride + j];
        float* o = out + (qi * P + j) * 3;
        o[0] = __fdiv_rn(__fsub_rn(raw[i * 3], qx), radius);
        o[1] = __fdiv_rn(__fsub_rn(raw[i * 3 + 1], qy), radius);
        o[2] = __fdiv_rn(__fsub_rn(raw[i * 3 + 2], qz), radius);
    }
}

extern "C" {

int pps_knn_f32(const float* pts, int64_t n, const float* query, int64_t m, int k, int64_t* out_idx, float* out_d2,
                void* stream) {
    if (n < 1 || m < 0 || k < 1 || k > 64 || k > n || n > 0x7fffffff) return PPS_ERR_ARG;
    if (m == 0) return PPS_OK;
    if (!pts || !query || !out_idx) return PPS_ERR_ARG;
    const int64_t ntask = (m + KNN_QW - 1) / KNN_QW;
    int64_t blocks = (ntask + KNN_WAVES - 1) / KNN_WAVES;
    int cus = pps_device_cu_count();
    if (cus <= 0) cus = 256;
    if (blocks > (int64_t)cus * 8) blocks = (int64_t)cus * 8;
    hipLaunchKernelGGL(knn_kernel, dim3((unsigned)blocks), dim3(KNN_WAVES * 64), 0, (hipStream_t)stream, pts, (int)n, query, m, k,
                       out_idx, out_d2);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_knn_blocked_f32(const float* pts_blocked, const int32_t* orig_idx, const float* bbox, int64_t nb, int64_t n,
                        const float* win_bbox, int64_t n_win, const float* query, int64_t m, int k, int64_t* out_idx, float* out_d2,
                        void* stream) {
    return pps_knn_blocked_groups_f32(pts_blocked, orig_idx, bbox, nb, n, win_bbox, n_win, nullptr, query, m, k, out_idx, out_d2, stream);
}

int pps_knn_blocked_groups_f32(const float* pts_blocked, const int32_t* orig_idx, const float* bbox, int64_t nb, int64_t n,
                               const float* win_bbox, int64_t n_win, const float* group_bbox, const float* query, int64_t m, int k,
                               int64_t* out_idx, float* out_d2, void* stream) {
    if (n < 1 || nb < 1 || nb * 64 < n || (nb - 1) * 64 >= n || m < 0 || k < 1 || k > 256 || k > n || nb > 0x1ffffff || n_win < 0)
        return PPS_ERR_ARG;
    if (m == 0) return PPS_OK;
    if (!pts_blocked || !orig_idx || !bbox || !query || !out_idx || (n_win > 0 && !win_bbox)) return PPS_ERR_ARG;
    const int64_t ntask = (m + KNN_QW - 1) / KNN_QW;
    int64_t blocks = (ntask + KNN_WAVES - 1) / KNN_WAVES;
    int cus = pps_device_cu_count();
    if (cus <= 0) cus = 256;
    if (blocks > (int64_t)cus * 8) blocks = (int64_t)cus * 8;
    const dim3 grid((unsigned)blocks), block(KNN_WAVES * 64);
    hipStream_t st = (hipStream_t)stream;
    const int r = (k + 63) / 64;
#define PPS_KNN_LAUNCH(R) hipLaunchKernelGGL(knn_blocked_kernel<R>, grid, block, 0, st, pts_blocked, orig_idx, bbox, (int)nb, win_bbox, \
                                             (int)n_win, group_bbox, query, m, k, out_idx, out_d2)
    if (r == 1) PPS_KNN_LAUNCH(1);
    else if (r == 2) PPS_KNN_LAUNCH(2);
    else if (r == 3) PPS_KNN_LAUNCH(3);
    else PPS_KNN_LAUNCH(4);
#undef PPS_KNN_LAUNCH
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_knn_blocked_batch_f32(int nkinds, int64_t nclouds, const float* const* pts_blocked, const int32_t* const* orig_idx,
                              const float* const* bbox, const int64_t* nb, const float* const* win_bbox, const int64_t* n_win,
                              const float* const* query, const int32_t* const* q_orig, const int64_t* query_stride, const int64_t* m,
                              const int* k, int64_t* const* out_idx, void* stream) {
    if (nkinds < 1 || nkinds > KBB_MAX || nclouds < 1 || !pts_blocked || !orig_idx || !bbox || !nb || !win_bbox || !n_win || !query || !q_orig ||
        !query_stride || !m || !k || !out_idx)
        return PPS_ERR_ARG;
    KbbArgs a;
    int64_t total = 0;
    for (int t = 0; t < nkinds; ++t) {
        if (!pts_blocked[t] || !orig_idx[t] || !bbox[t] || !query[t] || !out_idx[t] || nb[t] < 1 || nb[t] > 0x1ffffff || m[t] < 1 || k[t] < 1 ||
            k[t] > 64 || n_win[t] < 0 || (n_win[t] > 0 && !win_bbox[t]) || query_stride[t] < m[t] * 3)
            return PPS_ERR_ARG;
        KbbKind& kd = a.kind[t];
        kd.pts = pts_blocked[t]; kd.orig = orig_idx[t]; kd.bbox = bbox[t]; kd.win = win_bbox[t] ? win_bbox[t] : bbox[t];
        kd.query = query[t]; kd.q_orig = q_orig[t]; kd.out = out_idx[t];
        kd.pts_stride = nb[t] * 64 * 3; kd.orig_stride = nb[t] * 64; kd.bbox_stride = nb[t] * 6; kd.win_stride = n_win[t] * 6;
        kd.query_stride = query_stride[t]; kd.qorig_stride = query_stride[t] / 3; kd.out_stride = m[t] * k[t];
        kd.nb = (int)nb[t]; kd.n_win = (int)n_win[t]; kd.m = (int)m[t]; kd.k = k[t];
        kd.groups = (int)((m[t] + KNN_QW - 1) / KNN_QW);
        total += nclouds * kd.groups;
        if (total > 0x7fffffff) return PPS_ERR_ARG;
        kd.group_end = (int)total;
    }
    for (int t = nkinds; t < KBB_MAX; ++t) { a.kind[t] = a.kind[nkinds - 1]; a.kind[t].group_end = (int)total; }
    a.nkinds = nkinds;
    a.nclouds = (int)nclouds;
    int64_t blocks = (total + KNN_WAVES - 1) / KNN_WAVES;
    int cus = pps_device_cu_count();
    if (cus <= 0) cus = 256;
    if (blocks > (int64_t)cus * 8) blocks = (int64_t)cus * 8;
    hipLaunchKernelGGL(knn_blocked_batch_kernel, dim3((unsigned)blocks), dim3(KNN_WAVES * 64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_patch_normalize_f32(const float* raw, const float* query, const int64_t* idx, int64_t idx_stride, int64_t q, int p,
                            float* out, void* stream) {
    if (q < 0 || p < 1) return PPS_ERR_ARG;
    if (q == 0) return PPS_OK;
    if (!raw || !query || !idx || !out || idx_stride < p) return PPS_ERR_ARG;
    hipLaunchKernelGGL(patch_normalize_kernel, dim3((unsigned)((q + 3) / 4)), dim3(256), 0, (hipStream_t)stream, raw, query, idx,
                       idx_stride, q, p, out);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

}  // extern "C"
