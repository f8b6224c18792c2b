// micro-benchmark: what the f16 matrix pipe sustains on REAL (random) operands.  The chip clocks to its power budget (MI355X_MICROARCH.md, DVFS):
// a loop of nothing but MFMAs on zero operands and the same loop on random operands differ by the clock the chip holds.  Variants:
// 16x16x32 / 32x32x16, operands zero / random, 1 or 2 waves per SIMD, plus (LDS = 1) the A operands re-read from LDS every step like the decoder does.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_power_probe.hip -o /tmp/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int LDS>
__global__ __launch_bounds__(512, 2) void probe(const half8* __restrict__ src, float* out, int iters) {
    __shared__ half8 lds[8 * 64 * 8];                       // 64 KiB: 8 fragments per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = src[(i * 512 + threadIdx.x)]; lds[(wave * 8 + i) * 64 + lane] = a[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = src[((8 + i) * 512 + threadIdx.x)];
    __syncthreads();
    const half8* lp = lds + wave * 8 * 64 + lane;
    if (SHAPE == 16) {
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const half8 av = LDS ? lp[i * 64] : a[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b[j], acc[j], 0, 0, 0);
            }
            if (LDS) asm volatile("" ::: "memory");
        }
        f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
        out[blockIdx.x * 512 + threadIdx.x] = s.x + s.y + s.z + s.w;
    } else {
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const half8 av = LDS ? lp[i * 64] : a[i];
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, b[j], acc[j], 0, 0, 0);
            }
            if (LDS) asm volatile("" ::: "memory");
        }
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

template <int SHAPE, int LDS>
void run(const half8* src, float* out, int nth, const char* name) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<SHAPE, LDS>), dim3(grid), dim3(nth), 0, 0, src, out, 10);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<SHAPE, LDS>), dim3(grid), dim3(nth), 0, 0, src, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double nm = (double)iters * (SHAPE == 16 ? 32 : 16) * (nth / 64) * grid;
    const double tf = nm * (SHAPE == 16 ? 16384.0 : 32768.0) / (best * 1e-3) / 1e12;
    printf("%-70s %7.3f ms %7.1f TFLOP/s = %4.1f %% of 2500\n", name, best, tf, tf / 25.0);
    fflush(stdout);
}

int main() {
    const size_t n = 12 * 512 * 8;
    std::vector<uint16_t> h(n), z(n, 0);
    uint32_t st = 777u;
    for (size_t i = 0; i < n; ++i) { st = st * 1664525u + 1013904223u; h[i] = (uint16_t)(((st >> 31) << 15) | ((8 + ((st >> 20) & 3)) << 10) | ((st >> 8) & 0x3ff)); }
    half8 *dr, *dz; float* out;
    (void)hipMalloc(&dr, n * 2); (void)hipMalloc(&dz, n * 2); (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMemcpy(dr, h.data(), n * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(dz, z.data(), n * 2, hipMemcpyHostToDevice);
    run<16, 0>(dz, out, 512, "16x16x32 f16, zero operands, 2 waves/SIMD");
    run<16, 0>(dr, out, 512, "16x16x32 f16, random operands, 2 waves/SIMD");
    run<16, 0>(dr, out, 256, "16x16x32 f16, random operands, 1 wave/SIMD");
    run<32, 0>(dz, out, 512, "32x32x16 f16, zero operands, 2 waves/SIMD");
    run<32, 0>(dr, out, 512, "32x32x16 f16, random operands, 2 waves/SIMD");
    run<32, 0>(dr, out, 256, "32x32x16 f16, random operands, 1 wave/SIMD");
    run<16, 1>(dr, out, 512, "16x16x32 f16, random, A re-read from LDS (1 ds_read_b128 per 4 MFMA)");
    run<32, 1>(dr, out, 512, "32x32x16 f16, random, A re-read from LDS (1 ds_read_b128 per 2 MFMA)");
    run<16, 0>(dr, out, 512, "16x16x32 f16, random operands, 2 waves/SIMD (again, warm)");
    return 0;
}
