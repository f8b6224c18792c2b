// Device code shared by the FKAConv inference kernels (pps_fkaconv.hip) and the training kernels (pps_fka_train.hip):
// the geometry branch of source/base/nn.py:601-643 on the "16 lanes (one DPP row) = the K <= 16 neighbours of one
// support point" mapping, with the small per-layer parameters in one packed float array (`geo`).
#pragma once
#include "pps_common.h"

using namespace pps;

#define FK_TM 16                        // support points per workgroup (16 lanes each)
#define FK_NT 256

// packed small parameters of a layer ("geo" array, floats)
#define GEO_RADIUS 0
#define GEO_ALPHA 1
#define GEO_BETA 2
#define GEO_ACT 3                       // 1 relu, 2 silu
#define GEO_FC1 4                       // [16][3]
#define GEO_FC2 (GEO_FC1 + 48)          // [16][32]
#define GEO_FC3 (GEO_FC2 + 512)         // [16][32]
#define GEO_IN1W (GEO_FC3 + 512)
#define GEO_IN1B (GEO_IN1W + 16)
#define GEO_IN2W (GEO_IN1B + 16)
#define GEO_IN2B (GEO_IN2W + 16)
#define GEO_FLOATS (GEO_IN2B + 16)      // 1140

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == 2) return v / (1.f + __expf(-v));       // SiLU (ppsurf_model.py:49-50)
    return fmaxf(v, 0.f);
}

struct Geo {
    float dw;        // normalised distance weight of this neighbour (nn.py:619-624)
    float pn[3];     // neighbour offset / norm_radius (nn.py:601,616)
    bool valid;
};

__device__ __forceinline__ Geo geometry(const float* __restrict__ pts, const float* __restrict__ sup, const int64_t* __restrict__ idx,
                                        int64_t m, int64_t M, int j, int K, const float* geo) {
    Geo r;
    r.valid = (m < M) && (j < K);
    float d = 0.f;
    r.pn[0] = r.pn[1] = r.pn[2] = 0.f;
    if (r.valid) {
        const int64_t i = idx[m * K + j];
        const float px = pts[i * 3] - sup[m * 3], py = pts[i * 3 + 1] - sup[m * 3 + 1], pz = pts[i * 3 + 2] - sup[m * 3 + 2];
        d = sqrtf(px * px + py * py + pz * pz);
        const float rad = geo[GEO_RADIUS];
        r.pn[0] = px / rad; r.pn[1] = py / rad; r.pn[2] = pz / rad;
    }
    const float w = r.valid ? 1.f / (1.f + __expf(-(-geo[GEO_ALPHA] * d + geo[GEO_BETA]))) : 0.f;
    float s = row16_sum(w);
    s = s + (s == 0.f ? 1.f : 0.f) + 1e-6f;
    r.dw = w / s * (float)K;
    return r;
}

__device__ __forceinline__ void fc1_raw(const Geo& g, const float* geo, float (&o)[16]) {
#pragma unroll
    for (int t = 0; t < 16; ++t)
        o[t] = geo[GEO_FC1 + t * 3] * g.pn[0] + geo[GEO_FC1 + t * 3 + 1] * g.pn[1] + geo[GEO_FC1 + t * 3 + 2] * g.pn[2];
}

// act(IN(raw)) then [m ; max_j(m*dw)] -> fc (16x32)
__device__ __forceinline__ void norm_act_pool(float (&v)[16], const Geo& g, const float* geo, const float* stat /* [16][2] mean,rstd */,
                                              int wofs, int bofs, int K, float (&mp)[16]) {
    const int act = (int)geo[GEO_ACT];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        float x = v[t];
        if (K > 1) x = (x - stat[2 * t]) * stat[2 * t + 1] * geo[wofs + t] + geo[bofs + t];     // nn.py:627-630
        x = act_fn(x, act);
        v[t] = x;
        mp[t] = row16_max(g.valid ? x * g.dw : -INFINITY);
    }
}

__device__ __forceinline__ void fc32(const float (&a)[16], const float (&b)[16], const float* w /* [16][32] */, float (&o)[16]) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) s += w[t * 32 + c] * a[c];
#pragma unroll
        for (int c = 0; c < 16; ++c) s += w[t * 32 + 16 + c] * b[c];
        o[t] = s;
    }
}

