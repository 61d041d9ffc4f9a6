// How many single-wave workgroups run concurrently per CU on gfx950? (census by timing a spin kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) spin64(unsigned long long cycles, int* sink)
{
    __shared__ float pad[768];   // 3 KB like render_fwd
    pad[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
    if (pad[threadIdx.x] < 0) sink[0] = 1;
}
__global__ void __launch_bounds__(256) spin256(unsigned long long cycles, int* sink)
{
    __shared__ float pad[3072];
    pad[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
    if (pad[threadIdx.x] < 0) sink[0] = 1;
}
int main()
{
    int* sink; hipMalloc(&sink, 4);
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin64, 64, 0); printf("API: spin64 blocks/CU = %d\n", nb);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin256, 256, 0); printf("API: spin256 blocks/CU = %d\n", nb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned long long cyc = 2000000ull;  // ~20 ms at 100 MHz memtime
    for (int waves_per_cu : {8, 16, 24, 32, 40, 48, 64}) {
        float ms;
        hipEventRecord(a); hipLaunchKernelGGL(spin64, dim3(256 * waves_per_cu), dim3(64), 0, 0, cyc, sink); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("spin64  %2d blocks/CU requested: %.2f ms\n", waves_per_cu, ms);
    }
    for (int b4 : {2, 4, 8, 10}) {
        float ms;
        hipEventRecord(a); hipLaunchKernelGGL(spin256, dim3(256 * b4), dim3(256), 0, 0, cyc, sink); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("spin256 %2d blocks/CU requested: %.2f ms\n", b4, ms);
    }
    return 0;
}
