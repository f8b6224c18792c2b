// micro-benchmark: the steady state of the split-precision MFMA loops, piece by piece.  One chunk = 8 k-steps, a k-step = 4 ds_read_b128 (the
// hi / lo A fragments of two output blocks, requested one k-step ahead) + 6 v_mfma_f32_16x16x32_f16 in the dependency pattern of
// dense_blocks_f16x3.  Switches: BAR (one __syncthreads per chunk), DMA (0 none, 1 LDS-DMA of the next 32 KiB chunk as inline asm, 2 the hipcc
// builtin), NTH (512: 8 waves phase-locked, 256 x 2 workgroups per CU).  Prints time per chunk and the share of the f16 matrix peak.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/chunk_probe.hip -o tools/ubench/chunk_probe && tools/ubench/chunk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
#define CH4 2048
#define NCHUNK 18

template <int NTH, int BAR, int DMA, int NREADS>
__global__ __launch_bounds__(NTH, 2) void probe(const f32x4* __restrict__ w, float* out, int passes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* cur = (f32x4*)smem;
    f32x4* nxt = cur + CH4;
    constexpr int PER = CH4 / NTH;
    const int lane = threadIdx.x & 63;
    const int wave_base = threadIdx.x & ~63;
    unsigned lane_off = threadIdx.x * 16u;
    asm volatile("" : "+v"(lane_off));
    for (int i = threadIdx.x; i < 2 * CH4; i += NTH) cur[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    f32x4 m0 = {0, 0, 0, 0}, m1 = m0, c0 = m0, c1 = m0;
    half8 bh, bl;
#pragma unroll
    for (int j = 0; j < 8; ++j) { bh[j] = (_Float16)(0.001f * (lane + j)); bl[j] = (_Float16)(0.0001f * j); }
    for (int p = 0; p < passes; ++p) {
#pragma unroll 1
        for (int c = 0; c < NCHUNK; ++c) {
            const char* sbase = (const char*)(w + ((c + 1) % NCHUNK) * CH4);
            if (DMA == 1) {
                asm volatile("" : "+s"(sbase));
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(nxt + i * NTH + wave_base));
                    const char* piece = sbase + (size_t)(i * NTH * 16);
                    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_addr), "v"(lane_off), "s"(piece) : "memory", "m0");
                }
            } else if (DMA == 2) {
#pragma unroll
                for (int i = 0; i < PER; ++i)
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)(sbase + (size_t)(i * NTH * 16) + lane_off), (lds_ptr_t)(uintptr_t)(nxt + i * NTH + wave_base), 16, 0, 0);
            }
            const half8* w0 = (const half8*)cur + lane;
            half8 ph0 = w0[0], pl0 = w0[64], ph1 = w0[8 * 128], pl1 = w0[8 * 128 + 64];
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const half8 ah0 = ph0, al0 = pl0, ah1 = ph1, al1 = pl1;
                if (kb + 1 < 8 && NREADS) {
                    ph0 = w0[(kb + 1) * 128]; pl0 = w0[(kb + 1) * 128 + 64];
                    ph1 = w0[(8 + kb + 1) * 128]; pl1 = w0[(8 + kb + 1) * 128 + 64];
                }
                m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh, m0, 0, 0, 0);
                m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh, m1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bl, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bl, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bh, c1, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (DMA == 1) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
            if (BAR) __syncthreads();
            if (DMA) { f32x4* t = cur; cur = nxt; nxt = t; }
        }
    }
    const f32x4 s = m0 + m1 + c0 + c1;
    out[blockIdx.x * NTH + threadIdx.x] = s.x + s.y + s.z + s.w;
}

// Variant: THREE waves per SIMD (12 per CU).  Two waves share one 16-row tile, each contracting over half of the input channels (4 k-steps of the
// chunk = 24 MFMAs, 16 fragment reads per wave), the partner's partial output pair goes through LDS after the chunk barrier.  170 VGPRs per wave.
template <int EXCH>
__global__ __launch_bounds__(768, 1) void probe3(const f32x4* __restrict__ w, float* out, int passes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* cur = (f32x4*)smem;
    f32x4* nxt = cur + CH4;
    f32x4* xch = nxt + CH4;                              // [12 waves][2][64 lanes]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned lane_off = threadIdx.x * 16u;
    asm volatile("" : "+v"(lane_off));
    const int wave_base = threadIdx.x & ~63;
    for (int i = threadIdx.x; i < 2 * CH4; i += 768) cur[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0};
    half8 bh, bl;
#pragma unroll
    for (int j = 0; j < 8; ++j) { bh[j] = (_Float16)(0.001f * (lane + j)); bl[j] = (_Float16)(0.0001f * j); }
    const int k0 = (wave & 1) * 4;                        // this wave's half of the k-steps
    for (int p = 0; p < passes; ++p) {
#pragma unroll 1
        for (int c = 0; c < NCHUNK; ++c) {
            const char* sbase = (const char*)(w + ((c + 1) % NCHUNK) * CH4);
            asm volatile("" : "+s"(sbase));
            f32x4 m0 = {0, 0, 0, 0}, m1 = m0, c0 = m0, c1 = m0;
            const half8* w0 = (const half8*)cur + lane;
            half8 ph0 = w0[k0 * 128], pl0 = w0[k0 * 128 + 64], ph1 = w0[(8 + k0) * 128], pl1 = w0[(8 + k0) * 128 + 64];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int kb = k0 + kk;
                const half8 ah0 = ph0, al0 = pl0, ah1 = ph1, al1 = pl1;
                if (kk + 1 < 4) {
                    ph0 = w0[(kb + 1) * 128]; pl0 = w0[(kb + 1) * 128 + 64];
                    ph1 = w0[(8 + kb + 1) * 128]; pl1 = w0[(8 + kb + 1) * 128 + 64];
                }
                m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh, m0, 0, 0, 0);
                m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh, m1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bl, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bl, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bh, c1, 0, 0, 0);
                if (wave < 8 && kk < 4) {                 // the copy of the next chunk: 4 pieces from each of the first 8 waves
                    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(nxt + kk * 512 + wave_base));
                    const char* piece = sbase + (size_t)(kk * 512 * 16);
                    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_addr), "v"(lane_off), "s"(piece) : "memory", "m0");
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x4 o0 = m0 + c0, o1 = m1 + c1;
            const bool owner = ((c >> 2) & 1) == (wave & 1);       // which wave of the pair keeps this chunk's output pair
            if (EXCH && !owner) { xch[(wave * 2) * 64 + lane] = o0; xch[(wave * 2 + 1) * 64 + lane] = o1; }
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
            __syncthreads();
            if (EXCH && owner) { o0 += xch[((wave ^ 1) * 2) * 64 + lane]; o1 += xch[((wave ^ 1) * 2 + 1) * 64 + lane]; }
            acc += o0 + o1;
            f32x4* t = cur; cur = nxt; nxt = t;
        }
    }
    out[blockIdx.x * 768 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int EXCH>
void run3(const f32x4* w, float* out, const char* name) {
    const int passes = 60, grid = 256, lds = 2 * CH4 * 16 + 12 * 2 * 64 * 16;
    hipFuncSetAttribute((const void*)probe3<EXCH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe3<EXCH>), dim3(grid), dim3(768), lds, 0, w, out, 2);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe3<EXCH>), dim3(grid), dim3(768), lds, 0, w, out, passes);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)passes * NCHUNK * 24.0 * 12 * 256;            // 12 waves per CU, 24 MFMAs per wave and chunk
    const double tflops = mfma * 16384.0 / (ms * 1e-3) / 1e12;
    printf("%-58s %6.3f ms  %7.1f ns/chunk  %6.1f TFLOP/s = %4.1f %% of 2500\n", name, ms, ms * 1e6 / (passes * NCHUNK), tflops, tflops / 25.0);
}

template <int NTH, int BAR, int DMA, int NREADS>
void run(const f32x4* w, float* out, const char* name) {
    const int passes = 60, wgs = 512 / NTH;
    const int grid = 256 * wgs, lds = 2 * CH4 * 16;
    hipFuncSetAttribute((const void*)probe<NTH, BAR, DMA, NREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NTH, BAR, DMA, NREADS>), dim3(grid), dim3(NTH), lds, 0, w, out, 2);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NTH, BAR, DMA, NREADS>), dim3(grid), dim3(NTH), lds, 0, w, out, passes);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)passes * NCHUNK * 48.0 * 8 * 256;            // per chip: 8 waves per CU
    const double tflops = mfma * 16384.0 / (ms * 1e-3) / 1e12;
    printf("%-58s %6.3f ms  %7.1f ns/chunk  %6.1f TFLOP/s = %4.1f %% of 2500\n", name, ms, ms * 1e6 / (passes * NCHUNK), tflops, tflops / 25.0);
}

int main() {
    f32x4* w; float* out;
    (void)hipMalloc(&w, NCHUNK * CH4 * 16);
    (void)hipMemset(w, 0, NCHUNK * CH4 * 16);
    (void)hipMalloc(&out, 512 * 512 * 4);
    run<512, 0, 0, 0>(w, out, "MFMAs only (no fragment reads, no barrier, no stream)");
    run<512, 0, 0, 1>(w, out, "+ fragment reads from LDS");
    run<512, 1, 0, 1>(w, out, "+ barrier per chunk");
    run<512, 1, 1, 1>(w, out, "+ LDS-DMA stream (inline asm), 8-wave workgroup");
    run<512, 1, 2, 1>(w, out, "+ LDS-DMA stream (builtin), 8-wave workgroup");
    run<256, 1, 1, 1>(w, out, "+ LDS-DMA stream (inline asm), 2 x 4-wave workgroups");
    run<512, 1, 1, 0>(w, out, "stream + barrier, no fragment reads");
    run<512, 0, 1, 1>(w, out, "stream (asm) + reads, NO barrier (racy: timing only)");
    run3<0>(w, out, "12 waves, K split over wave pairs: stream + barrier + reads");
    run3<1>(w, out, "12 waves, K split, partial pair exchanged through LDS");
    return 0;
}
