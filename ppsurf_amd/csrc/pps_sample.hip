// Support-point sampling and batched neighbourhood tables of one encoder pass, for gfx950.
//
// pps_voxel_sample_f32   replaces source/poco_data_loader.py:59-134 `sampling_quantized` for one cloud (torch_geometric
//                        RandomRotate + voxel_grid + consecutive_cluster in a Python loop with a host sync per round):
//                        ONE workgroup runs all rounds of a level with the voxel hash table, the alive / representative
//                        flags and the reductions in LDS.
// pps_knn_multi_f32      the 13 kNN tables of source/poco_data_loader.py:155-168 (13 kd-tree builds + queries on the CPU)
//                        in ONE launch of the exhaustive search of pps_knn.hip.
#include "pps_common.h"
#include "../../include/ppsurf_amd.h"

#define VS_NT 1024
#define VS_TABLE 16384                 // hash slots (power of two); the kernel accepts n <= VS_MAXN points
#define VS_MAXN 10240

__device__ __forceinline__ unsigned vs_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ int block_sum_int(int v, int* red) {
    // red: LDS [17]; returns the sum over the workgroup to every thread
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
    for (int i = 0; i < VS_NT / 64; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float block_min_float(float v, float* red) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v = fminf(v, __shfl_xor(v, s));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < VS_NT / 64; ++i) t = fminf(t, red[i]);
    return t;
}

// pts [n,3]; rots [nrot][3][9]: per round the THREE row-major axis rotations in application order (x, y, z), each applied as
// p' = M p in float32 exactly like the reference's `pos @ matrix.t()` (one fp32 GEMM per axis, K = 3 evaluated by the GEMM
// microkernel as fma(p2, m2, fma(p1, m1, p0*m0)); oracle/driver_oracle.py rotate_f32, pinned by tests/golden/sampling.npz);
// prio: optional uint32 [n] -- the last round keeps the representatives with the smallest priority (ties: lower index)
// instead of ranking them by an internal hash ("truncation given the permutation", poco_data_loader.py:123);
// out_ids int64 [target] ascending.
// One workgroup per cloud: blockIdx.x selects cloud b of a batch of equally sized clouds (pts + b*n*3, rots + b*nrot*27,
// prio + b*n, out_ids + b*target, seed + b).
__device__ __forceinline__ void vs_rotate3(const float* __restrict__ R, float& x, float& y, float& z) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float* M = R + 9 * a;
        const float nx = __fmaf_rn(z, M[2], __fmaf_rn(y, M[1], __fmul_rn(x, M[0])));
        const float ny = __fmaf_rn(z, M[5], __fmaf_rn(y, M[4], __fmul_rn(x, M[3])));
        const float nz = __fmaf_rn(z, M[8], __fmaf_rn(y, M[7], __fmul_rn(x, M[6])));
        x = nx; y = ny; z = nz;
    }
}

typedef unsigned long long vs_u64;
#define VS_EMPTY64 0xffffffffffffffffull
#define VS_REP_BITS 14                 // n <= VS_MAXN = 10240 < 2^14
#define VS_CELL_MAX 65535              // 16 bits per axis: sqrt(target) * 2^round cells at most; 50 * 2^10 < 65536

__global__ __launch_bounds__(VS_NT) void voxel_sample_kernel(const float* __restrict__ pts_all, int n, int target, float vox,
                                                             const float* __restrict__ rots_all, int nrot, unsigned seed0,
                                                             const unsigned* __restrict__ prio_all,
                                                             int64_t* __restrict__ out_ids_all, int* __restrict__ out_rounds_all) {
    const float* __restrict__ pts = pts_all + (size_t)blockIdx.x * n * 3;
    const float* __restrict__ rots = rots_all + (size_t)blockIdx.x * nrot * 27;
    const unsigned* __restrict__ prio = prio_all ? prio_all + (size_t)blockIdx.x * n : nullptr;
    int64_t* __restrict__ out_ids = out_ids_all + (size_t)blockIdx.x * target;
    int* __restrict__ out_rounds = out_rounds_all ? out_rounds_all + blockIdx.x : nullptr;
    const unsigned seed = seed0 + blockIdx.x * 0x9e3779b9u;
    __shared__ vs_u64 tslot[VS_TABLE];           // (voxel key << 14) | representative: one 64-bit atomicMax keeps the LARGEST index
    __shared__ unsigned char state[VS_MAXN];     // bit0 alive, bit1 representative of this round, bit2 selected
    __shared__ int red_i[VS_NT / 64 + 1];
    __shared__ float red_f[VS_NT / 64 + 1];
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += VS_NT) state[i] = 1;
    __syncthreads();
    if (!(vox > 0.f)) {
        // default voxel edge: bounding-box diagonal / sqrt(target)   (poco_data_loader.py:85-88), float32, no contraction
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {INFINITY, INFINITY, INFINITY};      // hi holds the minimum of -x
        for (int i = tid; i < n; i += VS_NT)
            for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], pts[3 * i + c]); hi[c] = fminf(hi[c], -pts[3 * i + c]); }
        float e[3];
        for (int c = 0; c < 3; ++c) e[c] = __fsub_rn(-block_min_float(hi[c], red_f), block_min_float(lo[c], red_f));
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(e[0], e[0]), __fmul_rn(e[1], e[1])), __fmul_rn(e[2], e[2]));
        vox = __fdiv_rn(__fsqrt_rn(d2), __fsqrt_rn((float)target));
    }
    int count = 0, rounds = 0;
    bool done = false;
    for (int r = 0; r < nrot && !done; ++r, ++rounds) {
        const float* R = rots + r * 27;
        // bounding-box minimum of the rotated remaining points (voxel_grid anchors its grid there)
        float mx = INFINITY, my = INFINITY, mz = INFINITY;
        for (int i = tid; i < n; i += VS_NT)
            if (state[i] & 1) {
                float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
                vs_rotate3(R, x, y, z);
                mx = fminf(mx, x); my = fminf(my, y); mz = fminf(mz, z);
            }
        mx = block_min_float(mx, red_f); my = block_min_float(my, red_f); mz = block_min_float(mz, red_f);
        for (int s = tid; s < VS_TABLE; s += VS_NT) tslot[s] = VS_EMPTY64;
        __syncthreads();
        // one representative per occupied voxel: the LARGEST point index, like consecutive_cluster's sequential scatter_ on
        // the CPU (torch_geometric; oracle/driver_oracle.py consecutive_representatives)
        for (int i = tid; i < n; i += VS_NT)
            if (state[i] & 1) {
                float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
                vs_rotate3(R, x, y, z);
                const int cx = min(VS_CELL_MAX, (int)__fdiv_rn(__fsub_rn(x, mx), vox));
                const int cy = min(VS_CELL_MAX, (int)__fdiv_rn(__fsub_rn(y, my), vox));
                const int cz = min(VS_CELL_MAX, (int)__fdiv_rn(__fsub_rn(z, mz), vox));
                const vs_u64 key = (vs_u64)cx | ((vs_u64)cy << 16) | ((vs_u64)cz << 32);
                const vs_u64 mine = (key << VS_REP_BITS) | (vs_u64)i;
                unsigned slot = vs_hash((unsigned)key ^ vs_hash((unsigned)(key >> 24))) & (VS_TABLE - 1);
                for (;;) {
                    vs_u64 old = *(volatile vs_u64*)&tslot[slot];
                    if (old == VS_EMPTY64) {
                        old = atomicCAS(&tslot[slot], VS_EMPTY64, mine);
                        if (old == VS_EMPTY64) break;
                    }
                    if ((old >> VS_REP_BITS) == key) { atomicMax(&tslot[slot], mine); break; }
                    slot = (slot + 1) & (VS_TABLE - 1);
                }
            }
        __syncthreads();
        int mine_n = 0;
        for (int s = tid; s < VS_TABLE; s += VS_NT)
            if (tslot[s] != VS_EMPTY64) { state[(unsigned)(tslot[s] & ((1u << VS_REP_BITS) - 1))] |= 2; ++mine_n; }
        const int nrep = block_sum_int(mine_n, red_i);
        if (count + nrep < target) {
            // take every representative, drop it from the pool, halve the voxel
            for (int i = tid; i < n; i += VS_NT)
                if (state[i] & 2) state[i] = 4;
            count += nrep;
            vox *= 0.5f;
            __syncthreads();
        } else {
            // last round: a uniformly random subset of the representatives (rank by priority -- given, or a per-point hash --
            // threshold by bisection)
            const int need = target - count;
#define VS_PRIO(i) (prio ? prio[i] : vs_hash(seed ^ (unsigned)((i) * 2654435761u)))
            unsigned lo = 0u, hi = 0xffffffffu;           // smallest T with |{rep : h <= T}| >= need
            while (lo < hi) {
                const unsigned mid = lo + ((hi - lo) >> 1);
                int c = 0;
                for (int i = tid; i < n; i += VS_NT)
                    if ((state[i] & 2) && VS_PRIO(i) <= mid) ++c;
                c = block_sum_int(c, red_i);
                if (c >= need) hi = mid; else lo = mid + 1;
            }
            int below = 0;
            for (int i = tid; i < n; i += VS_NT)
                if ((state[i] & 2) && VS_PRIO(i) < lo) ++below;
            below = block_sum_int(below, red_i);
            // priority ties at the threshold: lowest indices first.  Rank among the tied representatives by a serial scan of
            // thread 0 (ties are practically absent with hashed / permuted priorities; correctness matters, speed does not)
            __shared__ int tie_left;
            if (tid == 0) tie_left = need - below;
            __syncthreads();
            for (int i = tid; i < n; i += VS_NT)
                if (state[i] & 2) {
                    const unsigned h = VS_PRIO(i);
                    if (h < lo) state[i] = 4;
                    else if (h > lo) state[i] &= 1;
                }
            __syncthreads();
            if (tid == 0)
                for (int i = 0; i < n; ++i)
                    if ((state[i] & 2) && !(state[i] & 4)) {          // tied at the threshold, ascending index
                        if (tie_left > 0) { state[i] = 4; --tie_left; } else state[i] &= 1;
                    }
            __syncthreads();
#undef VS_PRIO
            count = target;
            done = true;
        }
    }
    // out of rotations without reaching the target (not expected): fill with the lowest remaining indices
    __shared__ int fill_left;
    if (tid == 0) fill_left = target - count;
    __syncthreads();
    if (!done)
        for (int i = tid; i < n; i += VS_NT)
            if ((state[i] & 1) && atomicSub(&fill_left, 1) > 0) state[i] = 4;
    __syncthreads();
    // compact the selected ids in ascending order (per-thread contiguous ranges + exclusive scan of the range counts)
    __shared__ int offs[VS_NT + 1];
    const int per = (n + VS_NT - 1) / VS_NT, b0 = tid * per, b1 = min(n, b0 + per);
    int c = 0;
    for (int i = b0; i < b1; ++i) c += (state[i] & 4) ? 1 : 0;
    offs[tid + 1] = c;
    if (tid == 0) offs[0] = 0;
    __syncthreads();
    if (tid == 0)
        for (int i = 1; i <= VS_NT; ++i) offs[i] += offs[i - 1];
    __syncthreads();
    int o = offs[tid];
    for (int i = b0; i < b1; ++i)
        if (state[i] & 4) { if (o < target) out_ids[o] = i; ++o; }
    if (tid == 0 && out_rounds) *out_rounds = rounds;
}

// ---------------------------------------------------------------------------------------------------------------
// The same procedure for clouds beyond the LDS tables of voxel_sample_kernel (n > VS_MAXN: `pps.py fit` with manifold_points above 10240,
// whole clouds): the voxel hash table (key + representative in two arrays: the representative no longer has to fit beside the key in one
// 64-bit word), the state bytes and the compaction counts live in a caller workspace in global memory (L2-resident: 250 000 points need 7 MB).
// Still ONE workgroup per cloud -- the rounds are sequential and every step is a barrier-separated pass over the points, so the kernel is
// latency-bound (a few milliseconds at 250 000 points); it exists so that no cloud size leaves the device for the torch-op loop.
// Selections are identical to voxel_sample_kernel / the oracle given the same rotations and priorities (tests/test_gpu_sampling.py).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(VS_NT) void voxel_sample_big_kernel(const float* __restrict__ pts, int n, int target, float vox,
                                                                 const float* __restrict__ rots, int nrot, unsigned seed,
                                                                 const unsigned* __restrict__ prio, int64_t* __restrict__ out_ids,
                                                                 int* __restrict__ out_rounds, vs_u64* __restrict__ tkey, int* __restrict__ trep,
                                                                 unsigned table_mask, unsigned char* __restrict__ state, int* __restrict__ offs) {
    __shared__ int red_i[VS_NT / 64 + 1];
    __shared__ float red_f[VS_NT / 64 + 1];
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += VS_NT) state[i] = 1;
    __syncthreads();
    if (!(vox > 0.f)) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {INFINITY, INFINITY, INFINITY};
        for (int i = tid; i < n; i += VS_NT)
            for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], pts[3 * i + c]); hi[c] = fminf(hi[c], -pts[3 * i + c]); }
        float e[3];
        for (int c = 0; c < 3; ++c) e[c] = __fsub_rn(-block_min_float(hi[c], red_f), block_min_float(lo[c], red_f));
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(e[0], e[0]), __fmul_rn(e[1], e[1])), __fmul_rn(e[2], e[2]));
        vox = __fdiv_rn(__fsqrt_rn(d2), __fsqrt_rn((float)target));
    }
    int count = 0, rounds = 0;
    bool done = false;
    for (int r = 0; r < nrot && !done; ++r, ++rounds) {
        const float* R = rots + r * 27;
        float mx = INFINITY, my = INFINITY, mz = INFINITY;
        for (int i = tid; i < n; i += VS_NT)
            if (state[i] & 1) {
                float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
                vs_rotate3(R, x, y, z);
                mx = fminf(mx, x); my = fminf(my, y); mz = fminf(mz, z);
            }
        mx = block_min_float(mx, red_f); my = block_min_float(my, red_f); mz = block_min_float(mz, red_f);
        for (unsigned s = tid; s <= table_mask; s += VS_NT) { tkey[s] = VS_EMPTY64; trep[s] = -1; }
        __threadfence_block();
        __syncthreads();
        for (int i = tid; i < n; i += VS_NT)
            if (state[i] & 1) {
                float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
                vs_rotate3(R, x, y, z);
                const int cx = min(VS_CELL_MAX, (int)__fdiv_rn(__fsub_rn(x, mx), vox));
                const int cy = min(VS_CELL_MAX, (int)__fdiv_rn(__fsub_rn(y, my), vox));
                const int cz = min(VS_CELL_MAX, (int)__fdiv_rn(__fsub_rn(z, mz), vox));
                const vs_u64 key = (vs_u64)cx | ((vs_u64)cy << 16) | ((vs_u64)cz << 32);
                unsigned slot = vs_hash((unsigned)key ^ vs_hash((unsigned)(key >> 24))) & table_mask;
                for (;;) {
                    vs_u64 old = atomicCAS(&tkey[slot], VS_EMPTY64, key);       // claim an empty slot or find the voxel's slot
                    if (old == VS_EMPTY64 || old == key) { atomicMax(&trep[slot], i); break; }     // the LARGEST index represents the voxel
                    slot = (slot + 1) & table_mask;
                }
            }
        __threadfence_block();
        __syncthreads();
        int mine_n = 0;
        for (unsigned s = tid; s <= table_mask; s += VS_NT)
            if (tkey[s] != VS_EMPTY64) { state[trep[s]] |= 2; ++mine_n; }       // distinct slots hold distinct representatives: no write conflicts
        const int nrep = block_sum_int(mine_n, red_i);
        if (count + nrep < target) {
            for (int i = tid; i < n; i += VS_NT)
                if (state[i] & 2) state[i] = 4;
            count += nrep;
            vox *= 0.5f;
            __syncthreads();
        } else {
            const int need = target - count;
#define VS_PRIO(i) (prio ? prio[i] : vs_hash(seed ^ (unsigned)((i) * 2654435761u)))
            unsigned lo = 0u, hi = 0xffffffffu;
            while (lo < hi) {
                const unsigned mid = lo + ((hi - lo) >> 1);
                int c = 0;
                for (int i = tid; i < n; i += VS_NT)
                    if ((state[i] & 2) && VS_PRIO(i) <= mid) ++c;
                c = block_sum_int(c, red_i);
                if (c >= need) hi = mid; else lo = mid + 1;
            }
            int below = 0;
            for (int i = tid; i < n; i += VS_NT)
                if ((state[i] & 2) && VS_PRIO(i) < lo) ++below;
            below = block_sum_int(below, red_i);
            __shared__ int tie_left;
            if (tid == 0) tie_left = need - below;
            __syncthreads();
            for (int i = tid; i < n; i += VS_NT)
                if (state[i] & 2) {
                    const unsigned h = VS_PRIO(i);
                    if (h < lo) state[i] = 4;
                    else if (h > lo) state[i] &= 1;
                }
            __syncthreads();
            {
                // ties (priority == lo) are taken in index order until `tie_left` are found: a block-wide prefix count over contiguous index
                // ranges instead of one thread walking all n state bytes
                const int tper = (n + VS_NT - 1) / VS_NT, t0 = min(n, tid * tper), t1 = min(n, t0 + tper);
                int c = 0;
                for (int i = t0; i < t1; ++i) c += ((state[i] & 2) && !(state[i] & 4)) ? 1 : 0;
                offs[tid + 1] = c;
                if (tid == 0) offs[0] = 0;
                __threadfence_block();
                __syncthreads();
                if (tid == 0)
                    for (int i = 1; i <= VS_NT; ++i) offs[i] += offs[i - 1];
                __threadfence_block();
                __syncthreads();
                int before = offs[tid];
                const int left = tie_left;
                for (int i = t0; i < t1; ++i)
                    if ((state[i] & 2) && !(state[i] & 4)) {
                        if (before < left) state[i] = 4; else state[i] &= 1;
                        ++before;
                    }
            }
            __syncthreads();
#undef VS_PRIO
            count = target;
            done = true;
        }
    }
    __shared__ int fill_left;
    if (tid == 0) fill_left = target - count;
    __syncthreads();
    if (!done)
        for (int i = tid; i < n; i += VS_NT)
            if ((state[i] & 1) && atomicSub(&fill_left, 1) > 0) state[i] = 4;
    __syncthreads();
    const int per = (n + VS_NT - 1) / VS_NT, b0 = min(n, tid * per), b1 = min(n, b0 + per);
    int c = 0;
    for (int i = b0; i < b1; ++i) c += (state[i] & 4) ? 1 : 0;
    offs[tid + 1] = c;
    if (tid == 0) offs[0] = 0;
    __threadfence_block();
    __syncthreads();
    if (tid == 0)
        for (int i = 1; i <= VS_NT; ++i) offs[i] += offs[i - 1];
    __threadfence_block();
    __syncthreads();
    int o = offs[tid];
    for (int i = b0; i < b1; ++i)
        if (state[i] & 4) { if (o < target) out_ids[o] = i; ++o; }
    if (tid == 0 && out_rounds) *out_rounds = rounds;
}

static unsigned vs_big_slots(int64_t n) {
    unsigned s = 1u << 14;
    while ((int64_t)s < 2 * n) s <<= 1;                 // load factor <= 0.5
    return s;
}

// ---------------------------------------------------------------------------------------------------------------
// batched exhaustive kNN (same selection code as knn_kernel of pps_knn.hip, one wave per 8 queries of some table)
// ---------------------------------------------------------------------------------------------------------------
#define KM_MAX_TASKS 64
struct KnnMultiArgs {
    const float* pts[KM_MAX_TASKS];
    const float* query[KM_MAX_TASKS];
    int64_t* out[KM_MAX_TASKS];
    int n[KM_MAX_TASKS], m[KM_MAX_TASKS], k[KM_MAX_TASKS];
    int group_end[KM_MAX_TASKS];       // inclusive prefix sum of ceil(m/8)
    int ntasks;
};

typedef unsigned long long u64;
__device__ __forceinline__ u64 km_shfl_xor(u64 v, int m) {
    return ((u64)(unsigned)__shfl_xor((int)(unsigned)(v >> 32), m) << 32) | (unsigned)__shfl_xor((int)(unsigned)v, m);
}
__device__ __forceinline__ u64 km_shfl(u64 v, int src) {
    return ((u64)(unsigned)__shfl((int)(unsigned)(v >> 32), src) << 32) | (unsigned)__shfl((int)(unsigned)v, src);
}
__device__ __forceinline__ u64 km_sort64(u64 key, int lane) {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
        for (int st = size >> 1; st > 0; st >>= 1) {
            const u64 other = pps::lane_xor_u64(key, st, lane);
            const bool take_min = (((lane & size) == 0) == ((lane & st) == 0));
            const u64 mn = key < other ? key : other, mx = key < other ? other : key;
            key = take_min ? mn : mx;
        }
    return key;
}
__device__ __forceinline__ u64 km_merge64(u64 key, int lane) {
#pragma unroll
    for (int st = 32; st > 0; st >>= 1) {
        const u64 other = pps::lane_xor_u64(key, st, lane);
        const u64 mn = key < other ? key : other, mx = key < other ? other : key;
        key = ((lane & st) == 0) ? mn : mx;
    }
    return key;
}
__device__ __forceinline__ u64 km_flush(u64 list, const u64* cand, int cnt, int lane) {
    while (cnt > 0) {
        const int c = cnt < 64 ? cnt : 64;
        u64 ck = (lane < c) ? cand[cnt - c + lane] : ~0ull;
        ck = km_sort64(ck, lane);
        const u64 rev = km_shfl(ck, 63 - lane);
        list = km_merge64(list < rev ? list : rev, lane);
        cnt -= c;
    }
    return list;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define KM_QW 8
#define KM_WAVES 4
#define KM_CAP 128
__global__ __launch_bounds__(KM_WAVES * 64) void knn_multi_kernel(const KnnMultiArgs a) {
    __shared__ u64 cand_all[KM_WAVES][KM_QW][KM_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int total = a.group_end[a.ntasks - 1];
    for (int grp = blockIdx.x * KM_WAVES + wave; grp < total; grp += gridDim.x * KM_WAVES) {
        int t = 0;
        while (grp >= a.group_end[t]) ++t;
        const int g0 = t == 0 ? 0 : a.group_end[t - 1];
        const float* __restrict__ pts = a.pts[t];
        const float* __restrict__ query = a.query[t];
        const int n = a.n[t], m = a.m[t], k = a.k[t];
        const int q0 = (grp - g0) * KM_QW;
        float qx[KM_QW], qy[KM_QW], qz[KM_QW], tau[KM_QW];
        u64 list[KM_QW];
        int cnt[KM_QW];
#pragma unroll
        for (int j = 0; j < KM_QW; ++j) {
            const int qq = (q0 + j < m) ? q0 + j : m - 1;
            qx[j] = __shfl(query[qq * 3], 0); qy[j] = __shfl(query[qq * 3 + 1], 0); qz[j] = __shfl(query[qq * 3 + 2], 0);
            tau[j] = INFINITY; list[j] = ~0ull; cnt[j] = 0;
        }
        for (int base = 0; base < n; base += 64) {
            const int p = base + lane;
            const bool pv = p < n;
            const int pc = pv ? p : n - 1;
            const float px = pts[3 * pc], py = pts[3 * pc + 1], pz = pts[3 * pc + 2];
            // squared distances of the wave's 8 queries to this lane's point, TWO queries per instruction (v_pk_add_f32 / v_pk_mul_f32: the
            // scan, not the merge networks, is 2/3 of this kernel for k = 16).  Same operations in the same order as the scalar form
            // ((dx dx + dy dy) + dz dz, every step rounded: the build has -ffp-contract=off), so the keys -- and the tables -- are unchanged.
            float d2s[KM_QW];
#pragma unroll
            for (int j = 0; j < KM_QW; j += 2) {
                const f32x2 dx = f32x2{qx[j], qx[j + 1]} - f32x2{px, px}, dy = f32x2{qy[j], qy[j + 1]} - f32x2{py, py},
                            dz = f32x2{qz[j], qz[j + 1]} - f32x2{pz, pz};
                const f32x2 d = (dx * dx + dy * dy) + dz * dz;
                d2s[j] = d.x;
                d2s[j + 1] = d.y;
            }
#pragma unroll
            for (int j = 0; j < KM_QW; ++j) {
                const float d2 = d2s[j];
                const bool pass = pv && (d2 < tau[j]);
                const u64 mask = __ballot(pass);
                if (mask != 0ull) {
                    u64* cand = cand_all[wave][j];
                    const int pos = cnt[j] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                    if (pass) cand[pos] = ((u64)__float_as_uint(d2) << 32) | (unsigned)p;
                    cnt[j] += __popcll(mask);
                    if (cnt[j] > KM_CAP - 64) {
                        list[j] = km_flush(list[j], cand + (cnt[j] - 64), 64, lane);      // the newest 64: ONE sorting network per flush (two on 65..128
                        cnt[j] -= 64;                                                     // entries before); the older ones wait in the buffer
                        tau[j] = __uint_as_float((unsigned)(km_shfl(list[j], k - 1) >> 32));
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < KM_QW; ++j) {
            list[j] = km_flush(list[j], cand_all[wave][j], cnt[j], lane);
            if (q0 + j < m && lane < k) a.out[t][(int64_t)(q0 + j) * k + lane] = (int64_t)(unsigned)(list[j] & 0xffffffffull);
        }
    }
}

extern "C" {

int pps_voxel_sample_max_points(void) { return VS_MAXN; }

int pps_voxel_sample_f32(const float* pts, int64_t n, int64_t target, float vox, const float* rots, int nrot, uint32_t seed,
                         const uint32_t* priority, int64_t* out_ids, int32_t* out_rounds, void* stream) {
    if (!pts || !rots || !out_ids || n < 2 || n > VS_MAXN || target < 1 || target >= n || nrot < 1) return PPS_ERR_ARG;
    hipLaunchKernelGGL(voxel_sample_kernel, dim3(1), dim3(VS_NT), 0, (hipStream_t)stream, pts, (int)n, (int)target, vox, rots, nrot, seed,
                       priority, out_ids, out_rounds);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

size_t pps_voxel_sample_large_ws_bytes(int64_t n) {
    if (n < 2) return 0;
    const size_t slots = vs_big_slots(n);
    return slots * 8 + slots * 4 + (((size_t)n + 15) & ~(size_t)15) + (VS_NT + 1 + 3) * 4;
}

int pps_voxel_sample_large_f32(const float* pts, int64_t n, int64_t target, float vox, const float* rots, int nrot, uint32_t seed,
                               const uint32_t* priority, int64_t* out_ids, int32_t* out_rounds, void* ws, size_t ws_bytes, void* stream) {
    if (!pts || !rots || !out_ids || !ws || n < 2 || n > 0x3fffffff || target < 1 || target >= n || nrot < 1) return PPS_ERR_ARG;
    if (ws_bytes < pps_voxel_sample_large_ws_bytes(n) || ((uintptr_t)ws & 7) != 0) return PPS_ERR_ARG;
    const size_t slots = vs_big_slots(n);
    vs_u64* tkey = (vs_u64*)ws;
    int* trep = (int*)(tkey + slots);
    unsigned char* state = (unsigned char*)(trep + slots);
    int* offs = (int*)(state + (((size_t)n + 15) & ~(size_t)15));
    hipLaunchKernelGGL(voxel_sample_big_kernel, dim3(1), dim3(VS_NT), 0, (hipStream_t)stream, pts, (int)n, (int)target, vox, rots, nrot, seed, priority,
                       out_ids, out_rounds, tkey, trep, (unsigned)(slots - 1), state, offs);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_voxel_sample_batch_f32(const float* pts, int64_t b, int64_t n, int64_t target, const float* rots, int nrot, uint32_t seed,
                               const uint32_t* priority, int64_t* out_ids, int32_t* out_rounds, void* stream) {
    if (b == 0) return PPS_OK;
    if (!pts || !rots || !out_ids || b < 0 || n < 2 || n > VS_MAXN || target < 1 || target >= n || nrot < 1) return PPS_ERR_ARG;
    hipLaunchKernelGGL(voxel_sample_kernel, dim3((unsigned)b), dim3(VS_NT), 0, (hipStream_t)stream, pts, (int)n, (int)target, -1.f, rots, nrot,
                       seed, priority, out_ids, out_rounds);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_knn_multi_f32(int ntasks, const float* const* pts, const int64_t* n, const float* const* query, const int64_t* m, const int* k,
                      int64_t* const* out_idx, void* stream) {
    if (ntasks < 1 || ntasks > KM_MAX_TASKS || !pts || !n || !query || !m || !k || !out_idx) return PPS_ERR_ARG;
    KnnMultiArgs a;
    int total = 0;
    for (int t = 0; t < ntasks; ++t) {
        if (!pts[t] || !query[t] || !out_idx[t] || n[t] < 1 || m[t] < 1 || k[t] < 1 || k[t] > 64 || k[t] > n[t] || n[t] > 0x7fffffff ||
            m[t] > 0x3fffffff)
            return PPS_ERR_ARG;
        a.pts[t] = pts[t]; a.query[t] = query[t]; a.out[t] = out_idx[t];
        a.n[t] = (int)n[t]; a.m[t] = (int)m[t]; a.k[t] = k[t];
        total += (int)((m[t] + KM_QW - 1) / KM_QW);
        a.group_end[t] = total;
    }
    for (int t = ntasks; t < KM_MAX_TASKS; ++t) { a.pts[t] = nullptr; a.query[t] = nullptr; a.out[t] = nullptr; a.n[t] = a.m[t] = a.k[t] = 0; a.group_end[t] = total; }
    a.ntasks = ntasks;
    int blocks = (total + KM_WAVES - 1) / KM_WAVES;
    int cus = pps_device_cu_count();
    if (cus <= 0) cus = 256;
    if (blocks > cus * 8) blocks = cus * 8;
    hipLaunchKernelGGL(knn_multi_kernel, dim3(blocks), dim3(KM_WAVES * 64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

}  // extern "C"
