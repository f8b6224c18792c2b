// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds u16 element indices of a row-major [32 rows][PITCH] image; every lane supplies the
// address of one 8-byte piece (4 consecutive elements of one row) and the four u16 it RECEIVES are dumped, so the lane -> element
// map of the transpose can be read off.   hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int PITCH = 64;      // elements per image row

__global__ void probe(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t img[32 * PITCH];
    for (int i = threadIdx.x; i < 32 * PITCH; i += 64) img[i] = (uint16_t)i;       // element value = row * PITCH + col
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 15, kg = lane >> 4;
    // hypothesis: the 16 lanes of group kg supply the [4 rows][16 cols] block at rows 4*kg.., cols 0..15: lane i -> row 4kg + (i >> 2), cols 4 (i & 3)..+3,
    // and lane i receives column i of the block: rows 4kg + 0..3
    const unsigned addr = (unsigned)(uintptr_t)(&img[(4 * kg + (i >> 2)) * PITCH + 4 * (i & 3)]);
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[lane * 4 + 0] = (uint16_t)(v.x & 0xffff);
    out[lane * 4 + 1] = (uint16_t)(v.x >> 16);
    out[lane * 4 + 2] = (uint16_t)(v.y & 0xffff);
    out[lane * 4 + 3] = (uint16_t)(v.y >> 16);
}

int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    uint16_t h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            printf(" (r%2d,c%2d)", h[l * 4 + j] / PITCH, h[l * 4 + j] % PITCH);
            if (h[l * 4 + j] != (4 * (l >> 4) + j) * PITCH + (l & 15)) ok = 0;
        }
        printf("\n");
    }
    printf("hypothesis (lane (i, kg) element j = image[4 kg + j][i]) %s\n", ok ? "HOLDS" : "FAILS");
    return 0;
}
