// micro-benchmark: how fast can a persistent workgroup pull an L2-resident weight image through LDS while its waves run f16 MFMAs on it?
// The split-precision interpolation kernel (interp_pool_f16x3_kernel) re-streams 576 KB of weights per pass; this probe isolates that
// stream: 18 chunks of 32 KiB, double buffered, one barrier per chunk, per chunk and wave NREAD ds_read_b128 (A fragments) feeding NMFMA
// v_mfma_f32_16x16x32_f16.  MODE 0: LDS-DMA (global_load_lds_dwordx4), 1: global_load_dwordx4 -> VGPR -> ds_write_b128, 2: LDS-DMA nt.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_probe.hip -o tools/ubench/stream_probe && tools/ubench/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
#define CH4 2048
#define NCHUNK 18

template <int NTH, int MODE, int NMFMA, int NREAD>
__global__ __launch_bounds__(NTH) void probe(const f32x4* __restrict__ w, float* out, int passes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* cur = (f32x4*)smem;
    f32x4* nxt = cur + CH4;
    constexpr int PER = CH4 / NTH;
    const int lane = threadIdx.x & 63;
    const int wave_base = threadIdx.x & ~63;
    unsigned lane_off = threadIdx.x * 16u;
    asm volatile("" : "+v"(lane_off));
    auto issue_dma = [&](const f32x4* src, f32x4* dst) {
#pragma unroll
        for (int i = 0; i < PER; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)((const char*)src + (size_t)(i * NTH * 16) + lane_off), (lds_ptr_t)(uintptr_t)(dst + i * NTH + wave_base), 16, 0,
                                             MODE == 2 ? 2 : 0);
    };
    f32x4 stage[PER];
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < PER; ++i) cur[i * NTH + threadIdx.x] = w[i * NTH + threadIdx.x];
    } else {
        issue_dma(w, cur);
    }
    __syncthreads();
    f32x4 acc[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[c] = f32x4{0, 0, 0, 0};
    half8 bop;
#pragma unroll
    for (int j = 0; j < 8; ++j) bop[j] = (_Float16)(0.001f * (lane + j));
    for (int p = 0; p < passes; ++p) {
#pragma unroll 1
        for (int c = 0; c < NCHUNK; ++c) {
            const f32x4* gnext = w + ((c + 1) % NCHUNK) * CH4;
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < PER; ++i) stage[i] = gnext[i * NTH + threadIdx.x];
            } else {
                issue_dma(gnext, nxt);
            }
            if constexpr (NREAD > 0) {
                half8 a[NREAD > 0 ? NREAD : 1];
#pragma unroll
                for (int i = 0; i < NREAD; ++i) a[i] = ((const half8*)cur)[i * 64 + lane];
#pragma unroll
                for (int m = 0; m < NMFMA; ++m) acc[m % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m % NREAD], bop, acc[m % 6], 0, 0, 0);
            }
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < PER; ++i) nxt[i * NTH + threadIdx.x] = stage[i];
            }
            __syncthreads();
            f32x4* t = cur; cur = nxt; nxt = t;
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * NTH + threadIdx.x] = s + cur[threadIdx.x].x;
}

template <int NTH, int MODE, int NMFMA, int NREAD>
void run(int wg_per_cu, const f32x4* w, float* out, const char* name) {
    const int passes = 60;
    const int grid = 256 * wg_per_cu;
    const int lds = 2 * CH4 * 16;
    hipFuncSetAttribute((const void*)probe<NTH, MODE, NMFMA, NREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NTH, MODE, NMFMA, NREAD>), dim3(grid), dim3(NTH), lds, 0, w, out, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NTH, MODE, NMFMA, NREAD>), dim3(grid), dim3(NTH), lds, 0, w, out, passes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * passes * NCHUNK * CH4 * 16.0;
    const double mfma_cycles = (double)passes * NCHUNK * NMFMA * 16.0 * (NTH / 64) * wg_per_cu / 4.0;       // per SIMD, 16 cycles per 16x16x32 f16
    printf("%-46s NTH=%3d wg/CU=%d mfma/chunk=%3d reads=%2d : %6.3f ms  %5.2f TB/s  %5.1f B/clk/CU  us/pass %.2f  MFMA-bound %.3f ms (%.0f%%)\n", name, NTH, wg_per_cu, NMFMA,
           NREAD, ms, bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.4e9), ms * 1e3 / passes, mfma_cycles / 2.4e6, 100.0 * mfma_cycles / 2.4e6 / ms);
}

int main() {
    f32x4* w; float* out;
    hipMalloc(&w, NCHUNK * CH4 * 16);
    hipMemset(w, 0, NCHUNK * CH4 * 16);
    hipMalloc(&out, 512 * 512 * 4);
    // stream alone
    run<512, 0, 0, 0>(1, w, out, "LDS-DMA, stream only");
    run<512, 2, 0, 0>(1, w, out, "LDS-DMA nt, stream only");
    run<512, 1, 0, 0>(1, w, out, "VGPR staged, stream only");
    run<256, 0, 0, 0>(2, w, out, "LDS-DMA, stream only, 2 WG/CU (2x stream)");
    run<256, 1, 0, 0>(2, w, out, "VGPR staged, stream only, 2 WG/CU");
    // the f16x3 interpolation kernel's ratio: per chunk and wave 32 fragment reads, 48 MFMAs (one 16-row tile)
    run<512, 0, 48, 32>(1, w, out, "LDS-DMA + MFMA (1 tile/wave, 8 waves)");
    run<512, 2, 48, 32>(1, w, out, "LDS-DMA nt + MFMA (1 tile/wave, 8 waves)");
    run<512, 1, 48, 32>(1, w, out, "VGPR staged + MFMA (1 tile/wave, 8 waves)");
    // more rows per pass: 2 / 3 tiles per wave (96 / 144 MFMAs per 32 reads), 8 or 4 waves per CU
    run<512, 0, 96, 32>(1, w, out, "LDS-DMA + MFMA (2 tiles/wave, 8 waves)");
    run<512, 1, 96, 32>(1, w, out, "VGPR staged + MFMA (2 tiles/wave, 8 waves)");
    run<256, 0, 144, 32>(1, w, out, "LDS-DMA + MFMA (3 tiles/wave, 4 waves)");
    run<256, 1, 144, 32>(1, w, out, "VGPR staged + MFMA (3 tiles/wave, 4 waves)");
    run<256, 0, 96, 32>(1, w, out, "LDS-DMA + MFMA (2 tiles/wave, 4 waves)");
    run<256, 0, 192, 32>(1, w, out, "LDS-DMA + MFMA (4 tiles/wave, 4 waves)");
    return 0;
}
