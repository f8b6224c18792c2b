// Marching Cubes on the device: iso-surface of the float64 occupancy volume (NaN = never evaluated) as an indexed triangle mesh.
//
// replaces: skimage.measure.marching_cubes(volume, level) called at source/poco_utils.py:95-96 (round 3 ran it as ~40 torch ops, 3.4 ms per
// R = 257 shape).  Semantics = ppsurf_amd/mcubes.py::marching_cubes (the numpy twin the tests compare with, vertex for vertex and face for face):
//   * a cube is evaluated when its 8 corners are finite and not all on one side of the level ("inside" = value > level);
//   * the triangle list of a cube comes from a table indexed by (corner pattern, one bit per AMBIGUOUS face); the bit is the asymptotic decider
//     (sign of the bilinear saddle value of the face, Nielson & Hamann 1991; the face test of Lewiner et al. 2003): the two cubes that share a face read
//     its four values in the same order and therefore agree -- no cracks;
//   * [round 5] INTERIOR ambiguity (Chernyaev's / Lewiner's cases 4, 6, 7, 10, 12, 13): rows whose loops may be joined THROUGH the cube carry a list
//     of candidates (tun_index / tun_cand, derived in ppsurf_amd/mcubes.py): for each, the interior test of Lewiner et al. 2003 (test_interior) in
//     its general form -- sweep a plane along a cube axis, g(t) = X(t) Y(t) - B(t) D(t) of the plane's four columns, interior maximum of g with
//     X, Y of the groups' sign and g > 0 -- decides whether the two loops are closed by a TUBE (the candidate's alternative table row) instead of
//     two discs (sweep_connected below: the same operations in the same order as mcubes.interior_sweep_connected);
//   * vertices are welded by GRID-EDGE KEY, not by position: edge (voxel o, axis a) <-> key 3 * linear(o) + a; the vertex numbering is the ascending
//     key order, followed by the (rare) extra vertices inside a cube (table entry 12: a loop that has no chord-free triangulation is closed by a fan
//     around the mean of its crossing points), in cube order.
// Four streaming passes over the grid (HBM/L2-bound, ~0.14 GB each at R = 257), two block-level prefix sums in between (done by the caller: 68k
// integers), no atomics, deterministic output order:
//   pps_mc_count_f64   pass 1: per block of 256 cubes the number of triangles and of centre vertices; marks the used grid edges (byte flags)
//                      pass 2: per block of 1024 edge flags the number of vertices
//   pps_mc_emit_f64    pass 3: vertex positions + the edge -> vertex index map
//                      pass 4: faces (and the centre vertices)
#include "pps_common.h"
#include "../../include/ppsurf_amd.h"

namespace {

#define MC_NT 256
#define MC_EPT 4                    // edge flags per thread in the vertex passes

struct McDims {
    int nx, ny, nz;
    int64_t ncubes, nedges;         // (nx-1)(ny-1)(nz-1), 3 nx ny nz
};

// cube-corner c has offset (c & 1, (c >> 1) & 1, (c >> 2) & 1); cube edge e: axis e / 4, origin corner = the e % 4-th corner without that axis bit
__constant__ unsigned char MC_EDGE_A[12] = {0, 2, 4, 6, 0, 1, 4, 5, 0, 1, 2, 3};     // lower corner of edge e
__constant__ unsigned char MC_EDGE_AX[12] = {0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2};
// faces in the order of mcubes._FACE_UV: (axis, side) = (0,0) (0,1) (1,0) (1,1) (2,0) (2,1); corners at (u,v) = (0,0) (1,0) (0,1) (1,1), u = axis+1, v = axis+2
__constant__ unsigned char MC_FACE_UV[6][4] = {{0, 2, 4, 6}, {1, 3, 5, 7}, {0, 4, 1, 5}, {2, 6, 3, 7}, {0, 1, 2, 3}, {4, 5, 6, 7}};

__device__ __forceinline__ int block_scan_exclusive(int v, int* lds, int& total) {
    // exclusive prefix sum over the 256 threads of the block (wave shuffles + one LDS step)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += lds[w];
    total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + x - v;
}

struct CubeInfo {
    int row;                         // table row (pattern * 64 + decision bits), -1: cube not evaluated
    int pattern;
    double v[8];
};

// one sweep of the interior test: plane orthogonal to axis ax, diagonal pair dg of its four columns; w = sg * (v - level)
__device__ __forceinline__ bool sweep_connected(const double* v, double level, int sign, int ax, int dg) {
    const int u = (ax + 1) % 3, w_ = (ax + 2) % 3;
    const double sg = sign ? 1.0 : -1.0;
    // columns at (u, v) = (0,0) (1,0) (0,1) (1,1); diagonal pair 0: X = col 0, Y = col 3, B = col 1, D = col 2; pair 1: X = 1, Y = 2, B = 0, D = 3
    const int cx = dg ? 1 : 0, cy = dg ? 2 : 3, cb = dg ? 0 : 1, cd = dg ? 3 : 2;
#define MC_COL0(c) ((((c) & 1) << u) | ((((c) >> 1) & 1) << w_))
#define MC_W(c) (sg * (v[c] - level))
    const double x0 = MC_W(MC_COL0(cx)), x1 = MC_W(MC_COL0(cx) | (1 << ax)), y0 = MC_W(MC_COL0(cy)), y1 = MC_W(MC_COL0(cy) | (1 << ax));
    const double b0 = MC_W(MC_COL0(cb)), b1 = MC_W(MC_COL0(cb) | (1 << ax)), d0 = MC_W(MC_COL0(cd)), d1 = MC_W(MC_COL0(cd) | (1 << ax));
#undef MC_W
#undef MC_COL0
    const double dx = x1 - x0, dy = y1 - y0, db = b1 - b0, dd = d1 - d0;
    const double g2 = dx * dy - db * dd;
    const double g1 = (y0 * dx + x0 * dy) - (d0 * db + b0 * dd);
    if (!(g2 < 0.0)) return false;
    const double t = -g1 / (2.0 * g2);
    if (!(t > 0.0 && t < 1.0)) return false;
    const double xt = x0 + dx * t, yt = y0 + dy * t, bt = b0 + db * t, dt = d0 + dd * t;
    return xt > 0.0 && yt > 0.0 && (xt * yt - bt * dt > 0.0);
}

__device__ __forceinline__ CubeInfo classify(const double* __restrict__ vol, const McDims& d, int64_t lc, double level, const unsigned char* __restrict__ amb,
                                             const int* __restrict__ tun_index, const int* __restrict__ tun_cand) {
    CubeInfo ci;
    ci.row = -1;
    ci.pattern = 0;
    if (lc >= d.ncubes) return ci;
    const int cz = (int)(lc % (d.nz - 1));
    const int64_t r = lc / (d.nz - 1);
    const int cy = (int)(r % (d.ny - 1)), cx = (int)(r / (d.ny - 1));
    bool finite = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const double val = vol[((int64_t)(cx + (c & 1)) * d.ny + (cy + ((c >> 1) & 1))) * d.nz + (cz + ((c >> 2) & 1))];
        ci.v[c] = val;
        finite = finite && (val == val);
        ci.pattern |= (val > level) ? (1 << c) : 0;
    }
    if (!finite || ci.pattern == 0 || ci.pattern == 255) return ci;
    int dec = 0;
    const int am = amb[ci.pattern];
    if (am) {
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const double a = ci.v[MC_FACE_UV[f][0]] - level, b = ci.v[MC_FACE_UV[f][1]] - level;
            const double c = ci.v[MC_FACE_UV[f][2]] - level, e = ci.v[MC_FACE_UV[f][3]] - level;
            const double det = a * e - b * c;                      // -ffp-contract=off: two rounded products, one rounded difference, like numpy
            const bool joined = (a > 0.0) ? (det > 0.0) : (det < 0.0);
            dec |= joined ? (1 << f) : 0;
        }
        dec &= am;
    }
    ci.row = ci.pattern * 64 + dec;
    const int cnt = tun_index[2 * ci.row + 1];
    if (cnt) {
        // interior ambiguity: the first candidate pair of loops that one of its sweeps finds connected gets its tube row
        const int first = tun_index[2 * ci.row];
        for (int k = 0; k < cnt; ++k) {
            const int sign = tun_cand[3 * (first + k)], mask = tun_cand[3 * (first + k) + 1];
            bool hit = false;
            for (int s = 0; s < 6; ++s)
                if ((mask >> s) & 1) hit = hit || sweep_connected(ci.v, level, sign, s >> 1, s & 1);
            if (hit) { ci.row = tun_cand[3 * (first + k) + 2]; break; }
        }
    }
    return ci;
}

// pass 1
__global__ __launch_bounds__(MC_NT) void mc_count_cubes_kernel(const double* __restrict__ vol, McDims d, double level, const signed char* __restrict__ tri,
                                                               int width, const unsigned char* __restrict__ ntri, const unsigned char* __restrict__ amb,
                                                               const int* __restrict__ tun_index, const int* __restrict__ tun_cand,
                                                               unsigned char* __restrict__ flags, int* __restrict__ block_tris, int* __restrict__ block_centres) {
    __shared__ int lds[4];
    const int64_t lc = (int64_t)blockIdx.x * MC_NT + threadIdx.x;
    const CubeInfo ci = classify(vol, d, lc, level, amb, tun_index, tun_cand);
    int nt = 0, ncen = 0;
    if (ci.row >= 0) {
        nt = ntri[ci.row];
        ncen = (nt > 0 && tri[(int64_t)ci.row * width * 3] == 12) ? 1 : 0;      // a centre fan is listed first in its row (see below)
        const int cz = (int)(lc % (d.nz - 1));
        const int64_t r = lc / (d.nz - 1);
        const int cy = (int)(r % (d.ny - 1)), cx = (int)(r / (d.ny - 1));
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const int a = MC_EDGE_A[e], ax = MC_EDGE_AX[e];
            if (((ci.pattern >> a) & 1) != ((ci.pattern >> (a | (1 << ax))) & 1)) {
                const int64_t o = ((int64_t)(cx + (a & 1)) * d.ny + (cy + ((a >> 1) & 1))) * d.nz + (cz + ((a >> 2) & 1));
                flags[o * 3 + ax] = 1;                               // every cube around the edge stores the same byte
            }
        }
    }
    int tot;
    block_scan_exclusive(nt, lds, tot);
    if (threadIdx.x == 0) block_tris[blockIdx.x] = tot;
    block_scan_exclusive(ncen, lds, tot);
    if (threadIdx.x == 0) block_centres[blockIdx.x] = tot;
}

// pass 2
__global__ __launch_bounds__(MC_NT) void mc_count_verts_kernel(const unsigned char* __restrict__ flags, int64_t nedges, int* __restrict__ block_verts) {
    __shared__ int lds[4];
    const int64_t i0 = ((int64_t)blockIdx.x * MC_NT + threadIdx.x) * MC_EPT;
    int c = 0;
#pragma unroll
    for (int j = 0; j < MC_EPT; ++j) c += (i0 + j < nedges && flags[i0 + j]) ? 1 : 0;
    int tot;
    block_scan_exclusive(c, lds, tot);
    if (threadIdx.x == 0) block_verts[blockIdx.x] = tot;
}

// pass 3
__global__ __launch_bounds__(MC_NT) void mc_emit_verts_kernel(const double* __restrict__ vol, McDims d, double level, const unsigned char* __restrict__ flags,
                                                              const int64_t* __restrict__ vert_offset, int* __restrict__ vidx, double* __restrict__ verts) {
    __shared__ int lds[4];
    const int64_t i0 = ((int64_t)blockIdx.x * MC_NT + threadIdx.x) * MC_EPT;
    int c = 0;
    bool on[MC_EPT];
#pragma unroll
    for (int j = 0; j < MC_EPT; ++j) { on[j] = i0 + j < d.nedges && flags[i0 + j]; c += on[j] ? 1 : 0; }
    int tot;
    int64_t k = vert_offset[blockIdx.x] + block_scan_exclusive(c, lds, tot);
#pragma unroll
    for (int j = 0; j < MC_EPT; ++j) {
        if (!on[j]) continue;
        const int64_t e = i0 + j, o = e / 3;
        const int ax = (int)(e % 3);
        const int z = (int)(o % d.nz);
        const int64_t r = o / d.nz;
        const int y = (int)(r % d.ny), x = (int)(r / d.ny);
        const int64_t stride = ax == 0 ? (int64_t)d.ny * d.nz : (ax == 1 ? d.nz : 1);
        const double va = vol[o], vb = vol[o + stride];
        const double t = (level - va) / (vb - va);
        verts[k * 3 + 0] = (double)x + (ax == 0 ? t : 0.0);
        verts[k * 3 + 1] = (double)y + (ax == 1 ? t : 0.0);
        verts[k * 3 + 2] = (double)z + (ax == 2 ? t : 0.0);
        vidx[e] = (int)k;
        ++k;
    }
}

// pass 4
__global__ __launch_bounds__(MC_NT) void mc_emit_faces_kernel(const double* __restrict__ vol, McDims d, double level, const signed char* __restrict__ tri,
                                                              int width, const unsigned char* __restrict__ ntri, const unsigned char* __restrict__ amb,
                                                              const int* __restrict__ tun_index, const int* __restrict__ tun_cand,
                                                              const int* __restrict__ vidx, const int64_t* __restrict__ tri_offset,
                                                              const int64_t* __restrict__ centre_offset, int64_t n_edge_verts,
                                                              double* __restrict__ verts, int64_t* __restrict__ faces) {
    __shared__ int lds[4];
    const int64_t lc = (int64_t)blockIdx.x * MC_NT + threadIdx.x;
    const CubeInfo ci = classify(vol, d, lc, level, amb, tun_index, tun_cand);
    int nt = 0, ncen = 0;
    const signed char* row = nullptr;
    if (ci.row >= 0) {
        nt = ntri[ci.row];
        row = tri + (int64_t)ci.row * width * 3;
        ncen = (nt > 0 && row[0] == 12) ? 1 : 0;
    }
    int tot;
    const int64_t f0 = tri_offset[blockIdx.x] + block_scan_exclusive(nt, lds, tot);
    const int64_t cen = n_edge_verts + centre_offset[blockIdx.x] + block_scan_exclusive(ncen, lds, tot);
    if (nt == 0) return;
    const int cz = (int)(lc % (d.nz - 1));
    const int64_t r = lc / (d.nz - 1);
    const int cy = (int)(r % (d.ny - 1)), cx = (int)(r / (d.ny - 1));
    double sx = 0.0, sy = 0.0, sz = 0.0;
    int nfan = 0;
    for (int j = 0; j < nt; ++j) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int e = row[j * 3 + c];
            int64_t id;
            if (e == 12) {
                id = cen;
            } else {
                const int a = MC_EDGE_A[e], ax = MC_EDGE_AX[e];
                const int64_t o = ((int64_t)(cx + (a & 1)) * d.ny + (cy + ((a >> 1) & 1))) * d.nz + (cz + ((a >> 2) & 1));
                id = vidx[o * 3 + ax];
                if (c == 1 && row[j * 3] == 12) {
                    // second corner of a fan triangle: every vertex of the fan's loop exactly once -> the centre is their mean, summed in fan order
                    const double va = ci.v[a], vb = ci.v[a | (1 << ax)];
                    const double t = (level - va) / (vb - va);
                    sx += (double)(cx + (a & 1)) + (ax == 0 ? t : 0.0);
                    sy += (double)(cy + ((a >> 1) & 1)) + (ax == 1 ? t : 0.0);
                    sz += (double)(cz + ((a >> 2) & 1)) + (ax == 2 ? t : 0.0);
                    ++nfan;
                }
            }
            faces[(f0 + j) * 3 + c] = id;
        }
    }
    if (ncen) {
        verts[cen * 3 + 0] = sx / (double)nfan;
        verts[cen * 3 + 1] = sy / (double)nfan;
        verts[cen * 3 + 2] = sz / (double)nfan;
    }
}

bool dims_ok(int64_t nx, int64_t ny, int64_t nz) { return nx >= 2 && ny >= 2 && nz >= 2 && nx * ny * nz < (int64_t)700 * 1000 * 1000; }

McDims make_dims(int64_t nx, int64_t ny, int64_t nz) {
    McDims d;
    d.nx = (int)nx; d.ny = (int)ny; d.nz = (int)nz;
    d.ncubes = (nx - 1) * (ny - 1) * (nz - 1);
    d.nedges = 3 * nx * ny * nz;
    return d;
}

}  // namespace

extern "C" {

int64_t pps_mc_cube_blocks(int64_t nx, int64_t ny, int64_t nz) { return dims_ok(nx, ny, nz) ? ((nx - 1) * (ny - 1) * (nz - 1) + MC_NT - 1) / MC_NT : -1; }
int64_t pps_mc_edge_blocks(int64_t nx, int64_t ny, int64_t nz) { return dims_ok(nx, ny, nz) ? (3 * nx * ny * nz + MC_NT * MC_EPT - 1) / (MC_NT * MC_EPT) : -1; }

int pps_mc_count_f64(const double* vol, int64_t nx, int64_t ny, int64_t nz, double level, const int8_t* tri, int width, const uint8_t* ntri,
                     const uint8_t* amb, const int32_t* tun_index, const int32_t* tun_cand, uint8_t* edge_flags, int32_t* block_tris, int32_t* block_centres, int32_t* block_verts, void* stream) {
    if (!dims_ok(nx, ny, nz) || width < 1 || !vol || !tri || !ntri || !amb || !tun_index || !tun_cand || !edge_flags || !block_tris || !block_centres || !block_verts)
        return PPS_ERR_ARG;
    const McDims d = make_dims(nx, ny, nz);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(edge_flags, 0, (size_t)d.nedges, st) != hipSuccess) return PPS_ERR_LAUNCH;
    hipLaunchKernelGGL(mc_count_cubes_kernel, dim3((unsigned)pps_mc_cube_blocks(nx, ny, nz)), dim3(MC_NT), 0, st, vol, d, level, (const signed char*)tri, width,
                       ntri, amb, (const int*)tun_index, (const int*)tun_cand, edge_flags, block_tris, block_centres);
    hipLaunchKernelGGL(mc_count_verts_kernel, dim3((unsigned)pps_mc_edge_blocks(nx, ny, nz)), dim3(MC_NT), 0, st, (const unsigned char*)edge_flags, d.nedges,
                       block_verts);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

int pps_mc_emit_f64(const double* vol, int64_t nx, int64_t ny, int64_t nz, double level, const int8_t* tri, int width, const uint8_t* ntri,
                    const uint8_t* amb, const int32_t* tun_index, const int32_t* tun_cand, const uint8_t* edge_flags, const int64_t* tri_offset, const int64_t* centre_offset, const int64_t* vert_offset,
                    int64_t n_edge_verts, int32_t* vidx, double* verts, int64_t* faces, void* stream) {
    if (!dims_ok(nx, ny, nz) || width < 1 || n_edge_verts < 0 || n_edge_verts > 0x7fffffff) return PPS_ERR_ARG;
    if (!vol || !tri || !ntri || !amb || !tun_index || !tun_cand || !edge_flags || !tri_offset || !centre_offset || !vert_offset || !vidx || !verts || !faces) return PPS_ERR_ARG;
    const McDims d = make_dims(nx, ny, nz);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mc_emit_verts_kernel, dim3((unsigned)pps_mc_edge_blocks(nx, ny, nz)), dim3(MC_NT), 0, st, vol, d, level, (const unsigned char*)edge_flags,
                       vert_offset, vidx, verts);
    hipLaunchKernelGGL(mc_emit_faces_kernel, dim3((unsigned)pps_mc_cube_blocks(nx, ny, nz)), dim3(MC_NT), 0, st, vol, d, level, (const signed char*)tri, width,
                       ntri, amb, (const int*)tun_index, (const int*)tun_cand, (const int*)vidx, tri_offset, centre_offset, n_edge_verts, verts, faces);
    return hipGetLastError() == hipSuccess ? PPS_OK : PPS_ERR_LAUNCH;
}

}  // extern "C"
