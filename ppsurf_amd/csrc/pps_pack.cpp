// Host-side packing of dense layer weights into the A-operand order of v_mfma_f32_16x16x4_f32 used by the
// register-tile kernels (layout documented in pps_common.h).  Plain C++, no device code.
#include <cstddef>
#include <cstring>
#include <cstdint>
#include "../../include/ppsurf_amd.h"

static inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

extern "C" {

size_t pps_packed_dense_floats(int out, int in) {
    if (out <= 0 || in <= 0) return 0;
    return (size_t)pad_to(out, 32) * (size_t)pad_to(in, 16);
}

int pps_pack_dense_f32(const float* W, int out, int in, float* packed) {
    if (!W || !packed || out <= 0 || in <= 0) return 1;
    const int OB = pad_to(out, 32) / 16, KB = pad_to(in, 16) / 16;
    for (int ob = 0; ob < OB; ++ob)
        for (int kb = 0; kb < KB; ++kb)
            for (int l = 0; l < 64; ++l)
                for (int s = 0; s < 4; ++s) {
                    const int o = 16 * ob + (l & 15), c = 16 * kb + 4 * (l >> 4) + s;
                    packed[(((size_t)ob * KB + kb) * 64 + l) * 4 + s] = (o < out && c < in) ? W[(size_t)o * in + c] : 0.f;
                }
    return 0;
}

/* Split-precision A operands of v_mfma_f32_16x16x32_f16 (pps_common.h): per (ob, kb) 2 x 64 x 8 halfs,
 * packed[ob][kb][part][l][j] = part (0: hi = f16(W) round-to-nearest, 1: lo = f16(W - hi)) of
 * W[16 ob + (l & 15)][32 kb + 16 (j >> 2) + 4 (l >> 4) + (j & 3)]. */
size_t pps_packed_dense_f16x3_halfs(int out, int in) {
    if (out <= 0 || in <= 0) return 0;
    return (size_t)pad_to(out, 32) * (size_t)pad_to(in, 32) * 2;
}

int pps_pack_dense_f16x3(const float* W, int out, int in, uint16_t* packed) {
    if (!W || !packed || out <= 0 || in <= 0) return 1;
    const int OB = pad_to(out, 32) / 16, KB = pad_to(in, 32) / 32;
    for (int ob = 0; ob < OB; ++ob)
        for (int kb = 0; kb < KB; ++kb)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int o = 16 * ob + (l & 15), c = 32 * kb + 16 * (j >> 2) + 4 * (l >> 4) + (j & 3);
                    const float w = (o < out && c < in) ? W[(size_t)o * in + c] : 0.f;
                    const _Float16 hi = (_Float16)w;
                    const _Float16 lo = (_Float16)(w - (float)hi);
                    const size_t base = (((size_t)ob * KB + kb) * 2) * 64 * 8;
                    uint16_t hb, lb;
                    std::memcpy(&hb, &hi, 2);
                    std::memcpy(&lb, &lo, 2);
                    packed[base + (size_t)l * 8 + j] = hb;
                    packed[base + 64 * 8 + (size_t)l * 8 + j] = lb;
                }
    return 0;
}

size_t pps_packed_xyz_floats(int out) { return out <= 0 ? 0 : (size_t)pad_to(out, 16) * 4; }

int pps_pack_xyz_f32(const float* W, int out, float* packed) {
    if (!W || !packed || out <= 0) return 1;
    const int OB = pad_to(out, 16) / 16;
    for (int ob = 0; ob < OB; ++ob)
        for (int l = 0; l < 64; ++l) {
            const int o = 16 * ob + (l & 15), c = l >> 4;
            packed[ob * 64 + l] = (o < out && c < 3) ? W[o * 3 + c] : 0.f;
        }
    return 0;
}

}  // extern "C"
