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
#define VS_EMPTY 0xffffffffu

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

// pts [n,3]; rots [nrot][9] row-major rotation per round (p' = R p); out_ids int64 [target] ascending per round of selection
// One workgroup per cloud: blockIdx.x selects cloud b of a batch of equally sized clouds (pts + b*n*3, rots + b*nrot*9,
// out_ids + b*target, seed + b).
__global__ __launch_bounds__(VS_NT) void voxel_sample_kernel(const float* __restrict__ pts_all, int n, int target, float vox,
                                                             const float* __restrict__ rots_all, int nrot, unsigned seed0,
                                                             int64_t* __restrict__ out_ids_all, int* __restrict__ out_rounds_all) {
    const float* __restrict__ pts = pts_all + (size_t)blockIdx.x * n * 3;
    const float* __restrict__ rots = rots_all + (size_t)blockIdx.x * nrot * 9;
    int64_t* __restrict__ out_ids = out_ids_all + (size_t)blockIdx.x * target;
    int* __restrict__ out_rounds = out_rounds_all ? out_rounds_all + blockIdx.x : nullptr;
    const unsigned seed = seed0 + blockIdx.x * 0x9e3779b9u;
    __shared__ unsigned tkey[VS_TABLE];
    __shared__ unsigned trep[VS_TABLE];
    __shared__ unsigned char state[VS_MAXN];     // bit0 alive, bit1 representative of this round, bit2 selected
    __shared__ int red_i[VS_NT / 64 + 1];
    __shared__ float red_f[VS_NT / 64 + 1];
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += VS_NT) state[i] = 1;
    __syncthreads();
    if (!(vox > 0.f)) {
        // default voxel edge: bounding-box diagonal / sqrt(target)   (poco_data_loader.py:85-88)
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {INFINITY, INFINITY, INFINITY};      // hi holds the minimum of -x
        for (int i = tid; i < n; i += VS_NT)
            for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], pts[3 * i + c]); hi[c] = fminf(hi[c], -pts[3 * i + c]); }
        float d2 = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float e = -block_min_float(hi[c], red_f) - block_min_float(lo[c], red_f);
            d2 += e * e;
        }
        vox = sqrtf(d2) / sqrtf((float)target);
    }
    int count = 0, rounds = 0;
    bool done = false;
    for (int r = 0; r < nrot && !done; ++r, ++rounds) {
        const float* R = rots + r * 9;
        // bounding-box minimum of the rotated remaining points (voxel_grid anchors its grid there)
        float mx = INFINITY, my = INFINITY, mz = INFINITY;
        for (int i = tid; i < n; i += VS_NT)
            if (state[i] & 1) {
                const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
                mx = fminf(mx, R[0] * x + R[1] * y + R[2] * z);
                my = fminf(my, R[3] * x + R[4] * y + R[5] * z);
                mz = fminf(mz, R[6] * x + R[7] * y + R[8] * z);
            }
        mx = block_min_float(mx, red_f); my = block_min_float(my, red_f); mz = block_min_float(mz, red_f);
        for (int s = tid; s < VS_TABLE; s += VS_NT) { tkey[s] = VS_EMPTY; trep[s] = VS_EMPTY; }
        __syncthreads();
        // one representative (smallest index) per occupied voxel
        for (int i = tid; i < n; i += VS_NT)
            if (state[i] & 1) {
                const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
                const int cx = min(1023, (int)floorf((R[0] * x + R[1] * y + R[2] * z - mx) / vox));
                const int cy = min(1023, (int)floorf((R[3] * x + R[4] * y + R[5] * z - my) / vox));
                const int cz = min(1023, (int)floorf((R[6] * x + R[7] * y + R[8] * z - mz) / vox));
                const unsigned key = (unsigned)cx | ((unsigned)cy << 10) | ((unsigned)cz << 20);
                unsigned slot = vs_hash(key) & (VS_TABLE - 1);
                for (;;) {
                    const unsigned old = atomicCAS(&tkey[slot], VS_EMPTY, key);
                    if (old == VS_EMPTY || old == key) { atomicMin(&trep[slot], (unsigned)i); break; }
                    slot = (slot + 1) & (VS_TABLE - 1);
                }
            }
        __syncthreads();
        int mine = 0;
        for (int s = tid; s < VS_TABLE; s += VS_NT)
            if (tkey[s] != VS_EMPTY) { state[trep[s]] |= 2; ++mine; }
        const int nrep = block_sum_int(mine, red_i);
        if (count + nrep < target) {
            // take every representative, drop it from the pool, halve the voxel
            for (int i = tid; i < n; i += VS_NT)
                if (state[i] & 2) state[i] = 4;
            count += nrep;
            vox *= 0.5f;
            __syncthreads();
        } else {
            // last round: a uniformly random subset of the representatives (rank by a per-point hash, threshold by bisection)
            const int need = target - count;
            unsigned lo = 0u, hi = 0xffffffffu;           // smallest T with |{rep : h <= T}| >= need
            while (lo < hi) {
                const unsigned mid = lo + ((hi - lo) >> 1);
                int c = 0;
                for (int i = tid; i < n; i += VS_NT)
                    if ((state[i] & 2) && vs_hash(seed ^ (unsigned)(i * 2654435761u)) <= mid) ++c;
                c = block_sum_int(c, red_i);
                if (c >= need) hi = mid; else lo = mid + 1;
            }
            int below = 0;
            for (int i = tid; i < n; i += VS_NT)
                if ((state[i] & 2) && vs_hash(seed ^ (unsigned)(i * 2654435761u)) < lo) ++below;
            below = block_sum_int(below, red_i);
            // hash ties at the threshold (practically never more than one point): lowest indices first, serially
            __shared__ int tie_left;
            if (tid == 0) tie_left = need - below;
            __syncthreads();
            for (int i = tid; i < n; i += VS_NT)
                if (state[i] & 2) {
                    const unsigned h = vs_hash(seed ^ (unsigned)(i * 2654435761u));
                    if (h < lo) state[i] = 4;
                    else if (h == lo) { if (atomicSub(&tie_left, 1) > 0) state[i] = 4; else state[i] &= 1; }
                    else state[i] &= 1;
                }
            __syncthreads();
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
            const u64 other = km_shfl_xor(key, st);
            const bool take_min = (((lane & size) == 0) == ((lane & st) == 0));
            const u64 mn = key < other ? key : other, mx = key < other ? other : key;
            key = take_min ? mn : mx;
        }
    return key;
}
__device__ __forceinline__ u64 km_merge64(u64 key, int lane) {
#pragma unroll
    for (int st = 32; st > 0; st >>= 1) {
        const u64 other = km_shfl_xor(key, st);
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
#pragma unroll
            for (int j = 0; j < KM_QW; ++j) {
                const float dx = __fsub_rn(qx[j], px), dy = __fsub_rn(qy[j], py), dz = __fsub_rn(qz[j], pz);
                const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                const bool pass = pv && (d2 < tau[j]);
                const u64 mask = __ballot(pass);
                if (mask != 0ull) {
                    u64* cand = cand_all[wave][j];
                    const int pos = cnt[j] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                    if (pass) cand[pos] = ((u64)__float_as_uint(d2) << 32) | (unsigned)p;
                    cnt[j] += __popcll(mask);
                    if (cnt[j] > KM_CAP - 64) {
                        list[j] = km_flush(list[j], cand, cnt[j], lane);
                        cnt[j] = 0;
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
                         int64_t* out_ids, int32_t* out_rounds, void* stream) {
    if (!pts || !rots || !out_ids || n < 2 || n > VS_MAXN || target < 1 || target >= n || nrot < 1) return PPS_ERR_ARG;
    hipLaunchKernelGGL(voxel_sample_kernel, dim3(1), dim3(VS_NT), 0, (hipStream_t)stream, pts, (int)n, (int)target, vox, rots, nrot, seed,
                       out_ids, out_rounds);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_voxel_sample_batch_f32(const float* pts, int64_t b, int64_t n, int64_t target, const float* rots, int nrot, uint32_t seed,
                               int64_t* out_ids, int32_t* out_rounds, void* stream) {
    if (b == 0) return PPS_OK;
    if (!pts || !rots || !out_ids || b < 0 || n < 2 || n > VS_MAXN || target < 1 || target >= n || nrot < 1) return PPS_ERR_ARG;
    hipLaunchKernelGGL(voxel_sample_kernel, dim3((unsigned)b), dim3(VS_NT), 0, (hipStream_t)stream, pts, (int)n, (int)target, -1.f, rots, nrot,
                       seed, out_ids, out_rounds);
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
