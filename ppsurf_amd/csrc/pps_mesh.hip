// Mesh clean-up on the device (gfx950): the two steps of the reference's post-processing that are not Marching Cubes itself.
//
// replaces: source/base/mesh.py:7-38 as called from source/poco_utils.py:98-107, 169-174 -- trimesh's `merge_vertices(digits_vertex = 8)`,
// `remove_degenerate_faces`, `remove_duplicate_faces` and `remove_small_connected_components(num_faces = 6)` (split into face-connected components,
// keep those with more faces).  Until round 5 these were ~60 torch launches per clean-up (two sorts over all edges, six rounds of scatter-min label
// propagation); here:
//
//   pps_mesh_small_components   faces of components with <= k faces.  One hash-table pass links every face to the faces across its three edges
//                               (an edge's owners form a chain in arrival order: each (face, edge) gets the previous owner, and tells it about
//                               itself; no sort) and counts the owners of every edge; a second pass cuts the links of the edges that do not
//                               have exactly two (trimesh's face_adjacency joins across manifold edges only); then every face walks its own neighbourhood: a component of
//                               <= k faces is exhausted after visiting <= k faces, anything larger is left as soon as the (k+1)-th face shows up.
//                               No label propagation, no iteration to convergence, no atomics on floating point; the RESULT does not depend on
//                               the arrival order (connectivity is order-free).
//   pps_mesh_corner_weld        the Marching-Cubes kernels weld vertices by grid-edge key (pps_mc.hip); two vertices can still share a POSITION
//                               (to 8 digits) where their grid edges meet: on a grid corner.  Vertices within 10^-8 of a corner are entered into a
//                               hash table keyed by their rounded position (corner index + the -1/0/+1 unit of the 8th digit per axis: the key
//                               trimesh rounds to); every class is represented by its smallest vertex id.  -> remap [nv], the number of merged
//                               vertices, and the `hot` flags of the vertices something was merged into.
//   pps_mesh_face_filter        after remapping: drops degenerate faces and, among the faces that touch a hot vertex (only they can have become
//                               duplicates), all but the first face of every vertex triple (hash table on the triples themselves, exact).
// Hash tables: open addressing, linear probing, capacity a power of two >= 4 x entries (load <= 0.4), 64-bit keys claimed with atomicCAS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ppsurf_amd.h"

#define PPS_OK 0
#define PPS_ERR_ARG 1
#define PPS_ERR_LAUNCH 2

namespace {

typedef unsigned long long u64;
constexpr int MS_KMAX = 32;               // largest component size the filter can be asked for

__device__ __forceinline__ unsigned slot_of(u64 key, unsigned mask) { return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 32) & mask; }

// claims (or finds) the slot of `key` (never 0)
__device__ __forceinline__ unsigned hash_claim(u64* keys, unsigned mask, u64 key) {
    unsigned s = slot_of(key, mask);
    for (;;) {
        const u64 prev = atomicCAS(&keys[s], 0ull, key);
        if (prev == 0ull || prev == key) return s;
        s = (s + 1) & mask;
    }
}
__device__ __forceinline__ int hash_find(const u64* keys, unsigned mask, u64 key) {
    unsigned s = slot_of(key, mask);
    for (;;) {
        const u64 k = keys[s];
        if (k == key) return (int)s;
        if (k == 0ull) return -1;
        s = (s + 1) & mask;
    }
}

// ---- components ------------------------------------------------------------------------------------------------------------------------
// nbr[6 f + 2 e + 0] = the owner of edge e of face f that arrived before f (or -1), nbr[6 f + 2 e + 1] = the one that arrived after it (or -1)
__global__ __launch_bounds__(256) void mesh_edge_link_kernel(const int64_t* __restrict__ faces, int64_t nf, int64_t nv, u64* __restrict__ keys,
                                                            int* __restrict__ last, int* __restrict__ cnt, int* __restrict__ nbr, unsigned mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * nf) return;
    const int64_t f = i / 3;
    const int e = (int)(i - 3 * f);
    const int64_t a = faces[3 * f + e], b = faces[3 * f + (e == 2 ? 0 : e + 1)];
    const int64_t lo = a < b ? a : b, hi = a < b ? b : a;
    const unsigned s = hash_claim(keys, mask, (u64)lo * (u64)nv + (u64)hi + 1ull);
    atomicAdd(&cnt[s], 1);
    const int p = atomicExch(&last[s], (int)i);
    nbr[6 * f + 2 * e] = p < 0 ? -1 : p / 3;
    if (p >= 0) nbr[6 * (int64_t)(p / 3) + 2 * (p % 3) + 1] = (int)f;
}

// An edge joins its owners only if it has EXACTLY two: trimesh's face_adjacency (source/base/mesh.py:27) pairs the faces of the edges that occur
// twice; an edge shared by three or more faces (non-manifold: it can occur where centre fans and tubes of neighbouring cubes meet) joins nothing, so a
// small piece that hangs on the surface only by such an edge is a component of its own and is dropped like in the reference (rounds 3-5
// chained ALL owners).  The links of every other edge are cut again here.
__global__ __launch_bounds__(256) void mesh_edge_prune_kernel(const int64_t* __restrict__ faces, int64_t nf, int64_t nv, const u64* __restrict__ keys,
                                                             const int* __restrict__ cnt, int* __restrict__ nbr, unsigned mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * nf) return;
    const int64_t f = i / 3;
    const int e = (int)(i - 3 * f);
    const int64_t a = faces[3 * f + e], b = faces[3 * f + (e == 2 ? 0 : e + 1)];
    const int64_t lo = a < b ? a : b, hi = a < b ? b : a;
    const int s = hash_find(keys, mask, (u64)lo * (u64)nv + (u64)hi + 1ull);
    if (s >= 0 && cnt[s] != 2) { nbr[6 * f + 2 * e] = -1; nbr[6 * f + 2 * e + 1] = -1; }
}

__global__ __launch_bounds__(256) void mesh_small_kernel(const int* __restrict__ nbr, int64_t nf, int k, uint8_t* __restrict__ small) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= nf) return;
    int set[MS_KMAX + 1];
    int n = 1, head = 0;
    set[0] = (int)f;
    bool is_small = true;
    while (head < n && is_small) {
        const int g = set[head++];
        for (int s = 0; s < 6 && is_small; ++s) {
            const int h = nbr[6 * (int64_t)g + s];
            if (h < 0) continue;
            bool seen = false;
            for (int j = 0; j < n; ++j) seen |= set[j] == h;
            if (seen) continue;
            if (n == k) is_small = false;               // a (k+1)-th face: the component is larger than k
            else set[n++] = h;
        }
    }
    small[f] = is_small ? 1 : 0;
}

// ---- corner weld -----------------------------------------------------------------------------------------------------------------------
// key of a vertex near a grid corner: the position rounded to 10^-digits = corner index (19 bits per axis) + unit of the last digit in {-1, 0, +1}
__device__ __forceinline__ u64 corner_key(const double* v, double scale, double tol, int* err) {
    u64 key = 0;
    unsigned delta = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double c = rint(v[d]);
        if (!(fabs(v[d] - c) <= tol)) return 0ull;
        if (c < 0.0 || c > 524287.0) { *err = 1; return 0ull; }
        const double dq = rint(v[d] * scale) - c * scale;            // -1, 0 or +1 (exact: both terms are integers below 2^53)
        key = (key << 19) | (u64)(unsigned)(int)c;
        delta = delta * 3u + (unsigned)((int)dq + 1);
    }
    return ((key << 5) | delta) + 1ull;
}

__global__ __launch_bounds__(256) void mesh_corner_mark_kernel(const double* __restrict__ verts, int64_t nv, double scale, double tol,
                                                              u64* __restrict__ keys, int* __restrict__ rep, unsigned mask, int* __restrict__ err) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    const double p[3] = {verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]};
    const u64 key = corner_key(p, scale, tol, err);
    if (key == 0ull) return;
    const unsigned s = hash_claim(keys, mask, key);
    atomicMin(&rep[s], (int)v);
}

__global__ __launch_bounds__(256) void mesh_corner_remap_kernel(const double* __restrict__ verts, int64_t nv, double scale, double tol,
                                                               const u64* __restrict__ keys, const int* __restrict__ rep, unsigned mask,
                                                               int64_t* __restrict__ remap, uint8_t* __restrict__ hot, int* __restrict__ n_merged) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    const double p[3] = {verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]};
    int dummy = 0;
    const u64 key = corner_key(p, scale, tol, &dummy);
    int64_t r = v;
    if (key != 0ull) {
        const int s = hash_find(keys, mask, key);
        if (s >= 0) r = rep[s];
    }
    remap[v] = r;
    if (r != v) {
        hot[r] = 1;
        atomicAdd(n_merged, 1);
    }
}

// ---- faces after the weld --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sorted3(const int64_t* f, int64_t& a, int64_t& b, int64_t& c) {
    a = f[0]; b = f[1]; c = f[2];
    if (a > b) { const int64_t t = a; a = b; b = t; }
    if (b > c) { const int64_t t = b; b = c; c = t; }
    if (a > b) { const int64_t t = a; a = b; b = t; }
}
__device__ __forceinline__ unsigned triple_slot(int64_t a, int64_t b, int64_t c, unsigned mask) {
    return slot_of(((u64)a * 0x100000001B3ull) ^ ((u64)b * 0x9E3779B97F4A7C15ull) ^ ((u64)c + 0x7F4A7C15ull), mask);
}

// PASS 0: degenerate faces are dropped; faces around a hot vertex enter the table of triples: first[slot] = smallest face id with that triple
// PASS 1: such a face is kept iff it is that first one
template <int PASS>
__global__ __launch_bounds__(256) void mesh_face_filter_kernel(const int64_t* __restrict__ faces, int64_t nf, const uint8_t* __restrict__ hot,
                                                              int* __restrict__ first, unsigned mask, uint8_t* __restrict__ keep) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= nf) return;
    const int64_t* fv = faces + 3 * f;
    if (fv[0] == fv[1] || fv[1] == fv[2] || fv[0] == fv[2]) { if (PASS == 0) keep[f] = 0; return; }
    if (!(hot[fv[0]] | hot[fv[1]] | hot[fv[2]])) { if (PASS == 0) keep[f] = 1; return; }
    int64_t a, b, c;
    sorted3(fv, a, b, c);
    unsigned s = triple_slot(a, b, c, mask);
    for (;;) {
        int cur = first[s];
        if (PASS == 0 && cur < 0) {
            cur = atomicCAS(&first[s], -1, (int)f);
            if (cur < 0) { keep[f] = 1; return; }                   // claimed an empty slot
        }
        if (cur < 0) { keep[f] = 1; return; }                       // (PASS 1: cannot happen for an entered face)
        int64_t x, y, z;
        sorted3(faces + 3 * (int64_t)cur, x, y, z);
        if (x == a && y == b && z == c) {                           // the slot of this triple (all its faces share it, so its class never changes)
            if (PASS == 0) atomicMin(&first[s], (int)f);
            else keep[f] = first[s] == (int)f ? 1 : 0;
            return;
        }
        s = (s + 1) & mask;
    }
}

unsigned capacity_for(int64_t entries) {
    u64 cap = 1024;
    while (cap < (u64)entries * 4ull) cap <<= 1;
    return (unsigned)cap;
}
inline char* al256(char* p) { return (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255); }

}  // namespace

extern "C" {

size_t pps_mesh_components_ws_bytes(int64_t nf) {
    if (nf < 1 || nf > 100000000) return 0;
    const size_t cap = capacity_for(3 * nf);
    return 1024 + cap * (sizeof(u64) + 2 * sizeof(int)) + (size_t)nf * 6 * sizeof(int);
}

/* small [nf] = 1 for the faces of face-connected components (faces sharing an edge that has exactly two owners) with at most k faces
 * (1 <= k <= 32), else 0.
 * faces int64 [nf, 3] vertex ids below nv.  ws: pps_mesh_components_ws_bytes(nf) bytes. */
int pps_mesh_small_components(const int64_t* faces, int64_t nf, int64_t nv, int k, uint8_t* small, void* ws, void* stream) {
    if (nf < 0 || nv < 1 || k < 1 || k > MS_KMAX || nf > 100000000 || nv > 0x7fffffff) return PPS_ERR_ARG;      // (capacity 4 x 3 nf < 2^31 slots)
    if (nf == 0) return PPS_OK;
    if (!faces || !small || !ws) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned cap = capacity_for(3 * nf);
    u64* keys = (u64*)al256((char*)ws);
    int* cnt = (int*)(keys + cap);                                        // owners per edge slot
    int* last = cnt + cap;
    int* nbr = last + cap;
    if (hipMemsetAsync(keys, 0, (size_t)cap * (sizeof(u64) + sizeof(int)), st) != hipSuccess) return PPS_ERR_LAUNCH;                         // keys and cnt: 0
    if (hipMemsetAsync(last, 0xff, (size_t)cap * sizeof(int) + (size_t)nf * 6 * sizeof(int), st) != hipSuccess) return PPS_ERR_LAUNCH;     // last and nbr: -1
    hipLaunchKernelGGL(mesh_edge_link_kernel, dim3((unsigned)((3 * nf + 255) / 256)), dim3(256), 0, st, faces, nf, nv, keys, last, cnt, nbr, cap - 1);
    hipLaunchKernelGGL(mesh_edge_prune_kernel, dim3((unsigned)((3 * nf + 255) / 256)), dim3(256), 0, st, faces, nf, nv, (const u64*)keys, (const int*)cnt, nbr,
                       cap - 1);
    hipLaunchKernelGGL(mesh_small_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, (const int*)nbr, nf, k, small);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

size_t pps_mesh_weld_ws_bytes(int64_t nv) {
    if (nv < 1 || nv > 400000000) return 0;
    const size_t cap = capacity_for(nv);
    return 1024 + cap * (sizeof(u64) + sizeof(int));
}

/* Vertices within 10^-digits of a grid corner (verts float64 [nv, 3] in index space, coordinates in [0, 524287]) that share their position rounded
 * to `digits` digits are merged into the one with the smallest id: remap int64 [nv] (identity elsewhere), hot uint8 [nv] = 1 for vertices something
 * was merged into (zeroed here), counters int32 [2] = {number of merged vertices, 1 if a coordinate was outside the range}.  ws:
 * pps_mesh_weld_ws_bytes(nv) bytes. */
int pps_mesh_corner_weld(const double* verts, int64_t nv, int digits, int64_t* remap, uint8_t* hot, int* counters, void* ws, void* stream) {
    if (nv < 0 || nv > 400000000 || digits < 1 || digits > 12) return PPS_ERR_ARG;
    if (nv == 0) return PPS_OK;
    if (!verts || !remap || !hot || !counters || !ws) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    double scale = 1.0;
    for (int i = 0; i < digits; ++i) scale *= 10.0;
    const double tol = 1.0 / scale;
    const unsigned cap = capacity_for(nv);
    u64* keys = (u64*)al256((char*)ws);
    int* rep = (int*)(keys + cap);
    if (hipMemsetAsync(keys, 0, (size_t)cap * sizeof(u64), st) != hipSuccess) return PPS_ERR_LAUNCH;
    if (hipMemsetAsync(rep, 0x7f, (size_t)cap * sizeof(int), st) != hipSuccess) return PPS_ERR_LAUNCH;          // 0x7f7f7f7f: above every vertex id
    if (hipMemsetAsync(hot, 0, (size_t)nv, st) != hipSuccess || hipMemsetAsync(counters, 0, 2 * sizeof(int), st) != hipSuccess) return PPS_ERR_LAUNCH;
    const dim3 grid((unsigned)((nv + 255) / 256));
    hipLaunchKernelGGL(mesh_corner_mark_kernel, grid, dim3(256), 0, st, verts, nv, scale, tol, keys, rep, cap - 1, counters + 1);
    hipLaunchKernelGGL(mesh_corner_remap_kernel, grid, dim3(256), 0, st, verts, nv, scale, tol, (const u64*)keys, (const int*)rep, cap - 1, remap, hot,
                       counters);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

size_t pps_mesh_face_filter_ws_bytes(int64_t nf) {
    if (nf < 1 || nf > 400000000) return 0;
    return 1024 + (size_t)capacity_for(nf) * sizeof(int);
}

/* keep uint8 [nf]: 0 for degenerate faces (two equal vertex ids) and for every face that repeats the vertex set of a face with a smaller index, checked
 * among the faces that touch a `hot` vertex (uint8 [nv]); faces int64 [nf, 3] AFTER remapping.  ws: pps_mesh_face_filter_ws_bytes(nf) bytes. */
int pps_mesh_face_filter(const int64_t* faces, int64_t nf, const uint8_t* hot, uint8_t* keep, void* ws, void* stream) {
    if (nf < 0 || nf > 400000000) return PPS_ERR_ARG;
    if (nf == 0) return PPS_OK;
    if (!faces || !hot || !keep || !ws) return PPS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned cap = capacity_for(nf);
    int* first = (int*)al256((char*)ws);
    if (hipMemsetAsync(first, 0xff, (size_t)cap * sizeof(int), st) != hipSuccess) return PPS_ERR_LAUNCH;
    const dim3 grid((unsigned)((nf + 255) / 256));
    hipLaunchKernelGGL(mesh_face_filter_kernel<0>, grid, dim3(256), 0, st, faces, nf, hot, first, cap - 1, keep);
    hipLaunchKernelGGL(mesh_face_filter_kernel<1>, grid, dim3(256), 0, st, faces, nf, hot, first, cap - 1, keep);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

}  // extern "C"
