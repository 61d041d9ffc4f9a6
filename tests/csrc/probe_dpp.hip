// Probe of DPP / permlane data movement on gfx950 (documents the lane maps wave_reduce.h relies on).
#include <hip/hip_runtime.h>
#include <cstdio>
#define DPPI(old, src, ctrl, bank) __builtin_amdgcn_update_dpp((old), (src), (ctrl), 0xF, (bank), false)
__global__ void k(int* out)
{
    const int lane = threadIdx.x;
    out[0 * 64 + lane] = DPPI(-1, lane, 0x124, 0xF);   // row_ror:4
    out[1 * 64 + lane] = DPPI(-1, lane, 0x12C, 0xF);   // row_ror:12
    out[2 * 64 + lane] = DPPI(-1, lane, 0x128, 0xC);   // row_ror:8, banks 2,3
    out[3 * 64 + lane] = DPPI(-1, lane, 0x124, 0xA);   // row_ror:4, banks 1,3
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)lane, (unsigned)(100 + lane), false, false);
    out[4 * 64 + lane] = r[0]; out[5 * 64 + lane] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap((unsigned)lane, (unsigned)(100 + lane), false, false);
    out[6 * 64 + lane] = s[0]; out[7 * 64 + lane] = s[1];
    out[8 * 64 + lane] = DPPI(-1, lane, 0x104, 0xF);   // row_shl:4
    out[9 * 64 + lane] = DPPI(-1, lane, 0x114, 0xF);   // row_shr:4
}
int main()
{
    int *d, h[10 * 64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"row_ror:4", "row_ror:12", "row_ror:8 bank C", "row_ror:4 bank A", "swap32 vdst", "swap32 src", "swap16 vdst", "swap16 src", "row_shl:4", "row_shr:4"};
    for (int t = 0; t < 10; t++) { printf("%-18s:", names[t]); for (int l = 0; l < 64; l++) printf(" %d", h[t * 64 + l]); printf("\n"); }
    return 0;
}
