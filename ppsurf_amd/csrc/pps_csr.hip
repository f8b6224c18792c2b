// CSR of an id table of a fit batch (gfx950): "which (m, j) entries point at row n", entries of a row in ascending entry number.
//
// The backward pass of every neighbourhood gather is a scatter-add, done WITHOUT atomics on the values: each target row sums its
// contributions in the fixed order the CSR gives (pps_train.hip), so gradients are bit-reproducible.  The CSR itself is a stable
// counting sort of the entries by target row -- rows are small integers (< B * n), so no comparison / radix sort is needed:
//
//   count     flat[e] = ids[e] (+ item * rows_per_item, -1 -> 0 for the up-sampling tables); cnt[flat[e]] += 1     (integer atomics: exact)
//   tile sums + scan   offsets[r] = sum_{r' < r} cnt[r']      (two launches: sums of 2048-row tiles, then every tile scans itself on top of
//                      the sum of the tiles before it)
//   fill      tmp[offsets[r] + (--cnt[r])] = e                (arrival order inside a row is arbitrary; cnt is back at zero afterwards)
//   rank      order[offsets[r] + #{e' in row r : e' < e}] = e (every entry finds its place among the others of its row: ascending e,
//                                                              whatever the arrival order was -- the result is deterministic)
//
// Rows hold ~16 entries (K = 16 neighbour tables; 12.8 for the 64-NN projection table of 2000 queries on 10 000 points), so `rank`
// costs ~20 cached loads per entry.  Replaces torch.sort(stable) + torch.searchsorted (rocprim merge / radix sort passes,
// searchsorted, fill_reverse_indices: ~10 launches and 1.5 ms of the loader's queue per fit step for the 14 tables of a batch).
//
// replaces (reference, under autograd): the index_add backward of source/base/nn.py:655-674 `batch_gather` -- the reference
// scatters with atomics in whatever order the GPU runs them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ppsurf_amd.h"

namespace {

constexpr int TB = 256;
constexpr int TILE = 2048;       // rows per scan tile: 8 per thread

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 2; }
inline unsigned blocks_for(int64_t threads) { return (unsigned)((threads + TB - 1) / TB); }

__device__ __forceinline__ int64_t flat_of(const int64_t* __restrict__ ids, int64_t e, int64_t per_item, int64_t rows_per_item, int clamp) {
    int64_t v = ids[e];
    if (clamp && v < 0) v = 0;
    return per_item > 0 ? v + (e / per_item) * rows_per_item : v;
}

__global__ void __launch_bounds__(TB) csr_count_kernel(const int64_t* __restrict__ ids, int64_t entries, int64_t per_item, int64_t rows_per_item,
                                                       int64_t rows, int clamp, int64_t* __restrict__ flat, int* __restrict__ cnt) {
    const int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (e >= entries) return;
    const int64_t r = flat_of(ids, e, per_item, rows_per_item, clamp);
    if (flat) flat[e] = r;
    if ((uint64_t)r < (uint64_t)rows) atomicAdd(&cnt[r], 1);
}

// sum of a block's values (TB threads), result valid on every thread
__device__ __forceinline__ int64_t block_sum(int64_t v, int64_t* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    int64_t s = 0;
    for (int i = 0; i < TB / 64; ++i) s += sh[i];
    __syncthreads();
    return s;
}

__global__ void __launch_bounds__(TB) csr_tile_sum_kernel(const int* __restrict__ cnt, int64_t rows, int64_t* __restrict__ tile_sum) {
    __shared__ int64_t sh[TB / 64];
    const int64_t base = (int64_t)blockIdx.x * TILE;
    int64_t v = 0;
    for (int i = threadIdx.x; i < TILE; i += TB) {
        const int64_t r = base + i;
        if (r < rows) v += cnt[r];
    }
    const int64_t s = block_sum(v, sh);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = s;
}

// offsets[r] for the rows of this tile (+ offsets[rows] by the last tile).  Counts and offsets pass through LDS so that global memory sees unit-stride
// accesses (thread t owns the 8 CONSECUTIVE rows 8 t .. 8 t + 7 of the tile for the scan; read or written directly that is a 32- / 64-byte stride
// between lanes: 27 us per table in the first version of this kernel, profiles/round6_train_rocprof_summary.txt at commit 1b7b9f8)
__global__ void __launch_bounds__(TB) csr_scan_kernel(const int* __restrict__ cnt, int64_t rows, const int64_t* __restrict__ tile_sum,
                                                      int64_t* __restrict__ offsets) {
    __shared__ int64_t sh[TB / 64];
    __shared__ int64_t part[TB];
    __shared__ int cs[TILE];
    __shared__ int64_t os[TILE];
    const int64_t base = (int64_t)blockIdx.x * TILE;
    for (int i = threadIdx.x; i < TILE; i += TB) cs[i] = (base + i < rows) ? cnt[base + i] : 0;
    int64_t before = 0;
    for (int64_t t = threadIdx.x; t < (int64_t)blockIdx.x; t += TB) before += tile_sum[t];
    before = block_sum(before, sh);                         // (its barriers also publish cs)
    constexpr int PER = TILE / TB;
    int c[PER];
    int64_t mine = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        c[i] = cs[threadIdx.x * PER + i];
        mine += c[i];
    }
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < TB; o <<= 1) {                      // Hillis-Steele inclusive scan of the 256 thread sums
        const int64_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    int64_t run = before + part[threadIdx.x] - mine;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        os[threadIdx.x * PER + i] = run;
        run += c[i];
    }
    if (threadIdx.x == TB - 1 && blockIdx.x == gridDim.x - 1) offsets[rows] = run;      // rows past the end count 0: the last thread ends at the total
    __syncthreads();
    for (int i = threadIdx.x; i < TILE; i += TB)
        if (base + i < rows) offsets[base + i] = os[i];
}

__global__ void __launch_bounds__(TB) csr_fill_kernel(const int64_t* __restrict__ ids, int64_t entries, int64_t per_item, int64_t rows_per_item,
                                                      int64_t rows, int clamp, const int64_t* __restrict__ offsets, int* __restrict__ cnt,
                                                      int* __restrict__ tmp) {
    const int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (e >= entries) return;
    const int64_t r = flat_of(ids, e, per_item, rows_per_item, clamp);
    if ((uint64_t)r >= (uint64_t)rows) return;
    const int old = atomicSub(&cnt[r], 1);
    tmp[offsets[r] + old - 1] = (int)e;
}

__global__ void __launch_bounds__(TB) csr_rank_kernel(const int64_t* __restrict__ ids, int64_t per_item, int64_t rows_per_item, int clamp,
                                                      const int64_t* __restrict__ offsets, int64_t rows, const int* __restrict__ tmp,
                                                      int64_t* __restrict__ order) {
    const int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (p >= offsets[rows]) return;
    const int e = tmp[p];
    const int64_t r = flat_of(ids, e, per_item, rows_per_item, clamp);
    const int64_t lo = offsets[r], hi = offsets[r + 1];
    int64_t rank = 0;
    for (int64_t i = lo; i < hi; ++i) rank += tmp[i] < e;
    order[lo + rank] = e;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

size_t pps_csr_ws_bytes(int64_t entries, int64_t rows) {
    if (entries < 0 || rows < 0) return 0;
    const size_t tiles = (size_t)((rows + TILE - 1) / TILE);
    return align_up((size_t)rows * 4, 16) + align_up((size_t)entries * 4, 16) + align_up(tiles * 8, 16) + 16;
}

int pps_csr_build(const int64_t* ids, int64_t entries, int64_t per_item, int64_t rows_per_item, int64_t rows, int clamp_negative,
                  int64_t* flat, int64_t* order, int64_t* offsets, void* ws, size_t ws_bytes, void* stream) {
    if (entries < 0 || rows < 1 || per_item < 0 || rows_per_item < 0 || entries >= ((int64_t)1 << 31)) return 1;
    if (!offsets || !ws || ws_bytes < pps_csr_ws_bytes(entries, rows)) return 1;
    if (entries > 0 && (!ids || !order)) return 1;
    if (per_item > 0 && entries % per_item) return 1;
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    int* cnt = (int*)w;
    w += align_up((size_t)rows * 4, 16);
    int* tmp = (int*)w;
    w += align_up((size_t)entries * 4, 16);
    int64_t* tile_sum = (int64_t*)w;
    const unsigned tiles = (unsigned)((rows + TILE - 1) / TILE);
    if (hipMemsetAsync(cnt, 0, (size_t)rows * 4, st) != hipSuccess) return 2;
    if (entries > 0)
        csr_count_kernel<<<blocks_for(entries), TB, 0, st>>>(ids, entries, per_item, rows_per_item, rows, clamp_negative, flat, cnt);
    csr_tile_sum_kernel<<<tiles, TB, 0, st>>>(cnt, rows, tile_sum);
    csr_scan_kernel<<<tiles, TB, 0, st>>>(cnt, rows, tile_sum, offsets);
    if (entries > 0) {
        csr_fill_kernel<<<blocks_for(entries), TB, 0, st>>>(ids, entries, per_item, rows_per_item, rows, clamp_negative, offsets, cnt, tmp);
        csr_rank_kernel<<<blocks_for(entries), TB, 0, st>>>(ids, per_item, rows_per_item, clamp_negative, offsets, rows, tmp, order);
    }
    return launch_status();
}

}  // extern "C"
