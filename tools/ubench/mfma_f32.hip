// micro-benchmark: sustained v_mfma_f32_16x16x4_f32 rate vs number of independent accumulator chains and waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH>
void run(int wg_per_cu, const char* name) {
    float* out; hipMalloc(&out, 256 * 2048 * 4);
    int iters = 20000 / CH;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * wg_per_cu;
    hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * 4 * iters * 8 * CH * 2048.0;
    printf("%-28s chains=%d waves/SIMD=%d : %.1f TFLOP/s\n", name, CH, wg_per_cu, flop / ms / 1e9);
    hipFree(out);
}
int main() {
    run<1>(1, "1 chain"); run<2>(1, "2 chains"); run<4>(1, "4 chains");
    run<1>(2, "1 chain"); run<2>(2, "2 chains"); run<4>(2, "4 chains"); run<2>(4, "2 chains");
    return 0;
}
