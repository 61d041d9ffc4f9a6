// Can the matrix pipe take the backward's 12-value wave reduction off the VALU?  (docs/EXPERIMENTS.md section 7, round 3.)
// Every wave runs ITER rounds of { ~120 independent v_fma (the blend arithmetic's stand-in), then a reduction of 12 per-lane values over
// the 64 lanes }: (0) no reduction, (1) the butterfly transpose-reduce of ibgs_amd/csrc/wave_reduce.h, (2) thirteen v_mfma_f32_16x16x4_f32:
// twelve accumulate D[i][c] += sum_k V_c[16 k + i] (B = one-hot column c, shifted from c - 1 by a DPP row_shr:1), three adds fold D's four
// registers, one more MFMA with A = 1 sums the four lane rows -- lane l then holds the total of value l % 16.  Reports SIMD cycles per round at
// 8 waves per SIMD; the question is whether (2) - (0) < (1) - (0), i.e. whether the MFMAs overlap the other waves' VALU work.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -I ibgs_amd/csrc -o /tmp/pmr tests/csrc/probe_mfma_reduce.hip && /tmp/pmr
#include <hip/hip_runtime.h>
#include <cstdio>
#include "wave_reduce.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(64, 8) k(float* out, int iters, float seed, int nfma)
{
    const int lane = threadIdx.x;
    float acc[8];
    for (int i = 0; i < 8; i++) acc[i] = seed + lane * 0.001f + i;
    float total = 0.f;
    const float b0 = ((lane & 15) == 0) ? 1.0f : 0.0f;
    for (int it = 0; it < iters; it++) {
        for (int r = 0; r < nfma; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = fmaf(acc[i], 1.0000001f, 0.5f);
        }
        float v[12];
#pragma unroll
        for (int i = 0; i < 12; i++) v[i] = acc[i & 7] + (float)i;
        if (KIND == 1) {
            const float t = ibgs::wave_transpose_reduce12(v, lane);
            total += t;
        } else if (KIND == 2) {
            floatx4 d = {0.f, 0.f, 0.f, 0.f};
            float b = b0;
#pragma unroll
            for (int c = 0; c < 12; c++) {
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(v[c], b, d, 0, 0, 0);
                b = IBGS_DPP(0.f, b, 0x111 /* row_shr:1 */, 0xF);
            }
            const float t = (d[0] + d[1]) + (d[2] + d[3]);
            floatx4 z = {0.f, 0.f, 0.f, 0.f};
            const floatx4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, t, z, 0, 0, 0);
            total += d2[0];
        } else {
#pragma unroll
            for (int i = 0; i < 12; i++) total += v[i] * 1e-9f;
        }
    }
    out[blockIdx.x * 64 + lane] = total + acc[0];
}

// correctness of (2) against a plain sum
__global__ void check(float* out)
{
    const int lane = threadIdx.x;
    float v[12];
    for (int c = 0; c < 12; c++) v[c] = (float)((lane * 7 + c * 13) % 31) - 9.0f;
    floatx4 d = {0.f, 0.f, 0.f, 0.f};
    float b = ((lane & 15) == 0) ? 1.0f : 0.0f;
    for (int c = 0; c < 12; c++) { d = __builtin_amdgcn_mfma_f32_16x16x4f32(v[c], b, d, 0, 0, 0); b = IBGS_DPP(0.f, b, 0x111, 0xF); }
    const float t = (d[0] + d[1]) + (d[2] + d[3]);
    floatx4 z = {0.f, 0.f, 0.f, 0.f};
    const floatx4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, t, z, 0, 0, 0);
    out[lane] = d2[0];
}

int main()
{
    float* out; hipMalloc(&out, 1024 * 8 * 64 * 4 + 256);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, out);
    float h[64]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int c = 0; c < 12; c++) {
        float want = 0.f; for (int l = 0; l < 64; l++) want += (float)((l * 7 + c * 13) % 31) - 9.0f;
        for (int row = 0; row < 4; row++) if (h[row * 16 + c] != want) bad++;
    }
    printf("mfma reduce check: %s (value 3: %g)\n", bad ? "WRONG" : "ok", h[3]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int nfma : {5, 15, 40}) {
        float ms[3];
        for (int kind = 0; kind < 3; kind++) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(1024 * 8), dim3(64), 0, 0, out, iters, 1.0f, nfma);
                if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(1024 * 8), dim3(64), 0, 0, out, iters, 1.0f, nfma);
                if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(1024 * 8), dim3(64), 0, 0, out, iters, 1.0f, nfma);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1); if (rep && t < best) best = t;
            }
            ms[kind] = best;
        }
        // 8 waves per SIMD: cycles per round per WAVE-slot = ms * 2.4e6 / iters / 8
        auto cyc = [&](float m) { return m * 2.4e6 / iters / 8.0; };
        printf("%3d fma x8 per round: none %.0f  butterfly %.0f (+%.0f)  mfma %.0f (+%.0f)  SIMD cycles per round and wave\n", nfma,
               cyc(ms[0]), cyc(ms[1]), cyc(ms[1]) - cyc(ms[0]), cyc(ms[2]), cyc(ms[2]) - cyc(ms[0]));
    }
    return 0;
}
