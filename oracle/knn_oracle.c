/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product path
 * (ppsurf_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * CPU restatement of the exact k-nearest-neighbour search the reference performs through
 * pykdtree (source/poco_utils.py:257-273 `knn` -> source/base/proximity.py:84-89
 * `kdtree_query_oneshot` -> :40-81).  pykdtree (requirements.txt:17, ">=1.3") is a third-party
 * C/OpenMP kd-tree that is NOT under /root/reference; its published contract is: exact k nearest
 * data points per query, ascending distance, float32 arithmetic for float32 input.  Tie order is
 * traversal dependent and pinned by no reference test -> PARITY UNPINNED for ties.  This
 * restatement fixes the definition used by the build:
 *
 *   d2(q,p) = ((dx*dx + dy*dy) + dz*dz)  in IEEE float32, no FMA contraction   (compile with
 *   -ffp-contract=off), neighbours sorted by (d2, index) ascending.
 *
 * Layout: point-major float32 [n,3] / [m,3]; output int64 [m,k] (+ optional d2 float32 [m,k]).
 * k is clamped by the caller (poco_utils.py:259-260).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float d; int64_t i; } cand_t;

static inline int cand_less(float da, int64_t ia, float db, int64_t ib) {
    return (da < db) || (da == db && ia < ib);
}

/* sift-down on a max-heap ordered by (d, i) */
static void heap_sift(cand_t *h, int k, int pos) {
    for (;;) {
        int l = 2 * pos + 1, r = l + 1, big = pos;
        if (l < k && cand_less(h[big].d, h[big].i, h[l].d, h[l].i)) big = l;
        if (r < k && cand_less(h[big].d, h[big].i, h[r].d, h[r].i)) big = r;
        if (big == pos) return;
        cand_t t = h[pos]; h[pos] = h[big]; h[big] = t;
        pos = big;
    }
}

static int cand_cmp(const void *a, const void *b) {
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (cand_less(x->d, x->i, y->d, y->i)) return -1;
    if (cand_less(y->d, y->i, x->d, x->i)) return 1;
    return 0;
}

int pps_oracle_knn_f32(const float *pts, int64_t n, const float *query, int64_t m, int k,
                       int64_t *out_idx, float *out_d2) {
    if (k <= 0 || k > n) return 1;
#pragma omp parallel
    {
        cand_t *heap = (cand_t *)malloc(sizeof(cand_t) * (size_t)k);
#pragma omp for schedule(dynamic, 64)
        for (int64_t q = 0; q < m; ++q) {
            const float qx = query[3 * q], qy = query[3 * q + 1], qz = query[3 * q + 2];
            int filled = 0;
            for (int64_t p = 0; p < n; ++p) {
                const float dx = qx - pts[3 * p], dy = qy - pts[3 * p + 1], dz = qz - pts[3 * p + 2];
                const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const float s = xx + yy;
                const float d = s + zz;
                if (filled < k) {
                    heap[filled].d = d; heap[filled].i = p; ++filled;
                    if (filled == k)
                        for (int t = k / 2 - 1; t >= 0; --t) heap_sift(heap, k, t);
                } else if (cand_less(d, p, heap[0].d, heap[0].i)) {
                    heap[0].d = d; heap[0].i = p;
                    heap_sift(heap, k, 0);
                }
            }
            qsort(heap, (size_t)k, sizeof(cand_t), cand_cmp);
            for (int j = 0; j < k; ++j) {
                out_idx[q * k + j] = heap[j].i;
                if (out_d2) out_d2[q * k + j] = heap[j].d;
            }
        }
        free(heap);
    }
    return 0;
}
