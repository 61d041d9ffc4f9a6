// Stand-alone check of the wave64 bit-matrix transpose used by binning.hip.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/t tests/csrc/test_wave_bits.hip && /tmp/t
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../../ibgs_amd/csrc/wave_bits.h"

__global__ void k(const uint64_t* in, uint64_t* out, int reps)
{
    const int lane = threadIdx.x;
    const ibgs::BitTransposeConsts c = ibgs::bit_transpose_consts(lane);
    const uint64_t m = in[(blockIdx.x % 4096) * 64 + lane];
    uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    for (int r = 0; r < reps; r++) ibgs::wave_bit_transpose64(lo, hi, c);
    out[(blockIdx.x % 4096) * 64 + lane] = ((uint64_t)hi << 32) | lo;
}

int main()
{
    const int NB = 4096;
    uint64_t* h = (uint64_t*)malloc(NB * 64 * 8); uint64_t* r = (uint64_t*)malloc(NB * 64 * 8);
    srand(7);
    for (int i = 0; i < NB * 64; i++) {
        uint64_t v = 0; for (int k = 0; k < 4; k++) v = (v << 16) ^ (uint64_t)(rand() & 0xFFFF);
        if (i / 64 == 0) v = 1ull << (i % 64);                     // identity
        if (i / 64 == 1) v = (i % 64 == 5) ? ~0ull : 0ull;          // one full row
        if (i / 64 == 2) v = 1ull << 63;                            // one full column
        if (i / 64 % 3 == 0 && i / 64 > 2) v &= (uint64_t)rand() * 0x100000001ull;   // sparser
        h[i] = v;
    }
    uint64_t *d_in, *d_out;
    hipMalloc(&d_in, NB * 64 * 8); hipMalloc(&d_out, NB * 64 * 8);
    hipMemcpy(d_in, h, NB * 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, d_in, d_out, 1);
    hipMemcpy(r, d_out, NB * 64 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < NB; b++)
        for (int t = 0; t < 64; t++) {
            uint64_t want = 0;
            for (int l = 0; l < 64; l++) want |= ((h[b * 64 + l] >> t) & 1ull) << l;
            if (want != r[b * 64 + t]) { if (bad < 8) printf("block %d column %d: got %016llx want %016llx\n", b, t, (unsigned long long)r[b * 64 + t], (unsigned long long)want); bad++; }
        }
    // twice = identity
    hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, d_in, d_out, 2);
    hipMemcpy(r, d_out, NB * 64 * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < NB * 64; i++) if (r[i] != h[i]) bad++;
    // rate
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int REPS = 1000, BL = 256 * 32;
    hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, d_in, d_out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(BL), dim3(64), 0, 0, d_in, d_out, REPS | 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("transpose: %.1f cycles per wave per SIMD (8 waves/SIMD, 2.4 GHz)\n", ms * 1e-3 * 2.4e9 / ((double)REPS * BL / 1024.0));
    printf("wave_bit_transpose64: %s (%d bad)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
