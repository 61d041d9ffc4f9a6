// Stand-alone check of the wave64 butterfly transpose-reduces used by render_bwd.hip.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/t tests/csrc/test_wave_reduce.hip && /tmp/t
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../ibgs_amd/csrc/wave_reduce.h"

__global__ void k16(const float* in, float* out, int* col)
{
    const int lane = threadIdx.x;
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = in[lane * 16 + i];
    out[lane] = ibgs::wave_transpose_reduce16(v, lane);
    col[lane] = ibgs::reduce16_column(lane);
}
__global__ void k12(const float* in, float* out, int* col)
{
    const int lane = threadIdx.x;
    float v[12];
    for (int i = 0; i < 12; i++) v[i] = in[lane * 16 + i];
    out[lane] = ibgs::wave_transpose_reduce12(v, lane);
    col[lane] = ibgs::reduce12_column(lane);
}

static int check(const char* name, int nval, const float* h, const float* r, const int* c)
{
    int bad = 0, seen[16] = {0};
    for (int l = 0; l < 64; l++) {
        if (c[l] < 0) continue;
        if (c[l] >= nval) { bad++; continue; }
        seen[c[l]]++;
        double want = 0; for (int m = 0; m < 64; m++) want += h[m * 16 + c[l]];
        if (fabs(want - r[l]) > 1e-3) { bad++; if (bad < 8) printf("%s lane %d col %d: got %f want %f\n", name, l, c[l], r[l], want); }
    }
    for (int i = 0; i < nval; i++) if (seen[i] != 1) { bad++; printf("%s: column %d owned by %d lanes\n", name, i, seen[i]); }
    printf("%s: %s (%d bad)\n", name, bad ? "FAIL" : "OK", bad);
    return bad;
}

int main()
{
    float h[64 * 16], *d, *o, r[64];
    int *dc, c[64];
    for (int l = 0; l < 64; l++) for (int i = 0; i < 16; i++) h[l * 16 + i] = (float)((l * 7 + i * 13) % 11) + 0.25f * i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r)); hipMalloc(&dc, sizeof(c));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    int bad = 0;
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, d, o, dc);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost); hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    bad += check("wave_transpose_reduce16", 16, h, r, c);
    hipLaunchKernelGGL(k12, dim3(1), dim3(64), 0, 0, d, o, dc);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost); hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    bad += check("wave_transpose_reduce12", 12, h, r, c);
    return bad != 0;
}
