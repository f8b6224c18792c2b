#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ char smem[];
__global__ __launch_bounds__(256) void k(unsigned* out, int spin) {
    unsigned lds = __builtin_amdgcn_s_getreg((6) | (0 << 6) | ((8 - 1) << 11));
    unsigned hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
    smem[threadIdx.x] = 1;
    long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = lds; out[blockIdx.x * 4 + 1] = hwid; out[blockIdx.x * 4 + 2] = xcc; out[blockIdx.x*4+3] = (unsigned)(t0 & 0xffffffff); }
}
int main() {
    unsigned* d; hipMalloc(&d, 512 * 16);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 79104);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 79104, 0, d, 200000);
    unsigned h[2048]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int nz = 0; for (int b = 0; b < 512; ++b) nz += h[b * 4] != 0;
    printf("blocks with nonzero LDS base: %d of 512\n", nz);
    for (int b : {0, 1, 8, 9, 255, 256, 257, 264, 511}) {
        unsigned hw = h[b * 4 + 1];
        printf("block %3d lds_base=%3u xcc=%u cu=%u se=%u sh=%u simd=%u wave=%u t0=%u\n", b, h[b * 4], h[b * 4 + 2], (hw >> 8) & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 4) & 3, hw & 15, h[b*4+3]);
    }
    // which blocks share (xcc,se,sh,cu)?
    int same = 0;
    for (int a = 0; a < 256; ++a) { unsigned ka = (h[a*4+2] << 16) | (h[a*4+1] & 0xff00); unsigned kb = (h[(a+256)*4+2] << 16) | (h[(a+256)*4+1] & 0xff00); same += ka == kb; }
    printf("block b and b+256 on the same CU: %d of 256\n", same);
    return 0;
}
