// Stand-alone check of the wave64 butterfly transpose-reduce used by render_bwd.hip.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/t tests/csrc/test_wave_reduce.hip && /tmp/t
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define IBGS_TEST_REDUCE_ONLY 1
#include "../../ibgs_amd/csrc/wave_reduce.h"

__global__ void k(const float* in, float* out)
{
    const int lane = threadIdx.x;
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = in[lane * 16 + i];
    out[lane] = ibgs::wave_transpose_reduce16(v, lane);
}

int main()
{
    float h[64 * 16], *d, *o, r[64];
    for (int l = 0; l < 64; l++) for (int i = 0; i < 16; i++) h[l * 16 + i] = (float)((l * 7 + i * 13) % 11) + 0.25f * i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        double want = 0; for (int m = 0; m < 64; m++) want += h[m * 16 + (l >> 2)];
        if (fabs(want - r[l]) > 1e-3) { bad++; if (bad < 8) printf("lane %d: got %f want %f\n", l, r[l], want); }
    }
    printf("wave_transpose_reduce16: %s (%d bad lanes)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
