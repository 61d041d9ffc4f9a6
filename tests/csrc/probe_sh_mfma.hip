// Measures the north star's MFMA candidate instead of arguing it: SH degree 3 -> RGB
// (forward.cu:58-109: rgb[c] = sum_k basis_k(dir) * sh[k][c], 16 coefficients x 3 channels per Gaussian) for 64 Gaussians
// per wave, operands already in registers (the memory side is the same for both):
//   VALU  : lane = Gaussian, 48 v_fma_f32 per wave instruction stream (what preprocess.hip does);
//   MFMA  : v_mfma_f32_4x4x1_16B_f32, block = Gaussian (16 Gaussians per instruction), A = sh[k][0..3] (3 of 4 rows
//           used), B = basis_k replicated over the 4 columns, accumulated over the 16 coefficients: 16 MFMAs per 16
//           Gaussians = 64 per 64 Gaussians at 3/16 useful MACs (a different matrix per Gaussian: there is no operand
//           shared across Gaussians to contract over);
//   MFMA16: v_mfma_f32_16x16x4_f32 with a block-diagonal packing: rows = 16 (Gaussian, channel) pairs is not possible
//           either -- the contraction index k must carry the SAME B row for every output column, so each Gaussian still needs
//           its own instruction; measured as 4 Gaussians per instruction (k = 4 coefficients per step, 4 steps, 1 of 16
//           columns useful per Gaussian row block).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/psh tests/csrc/probe_sh_mfma.hip && /tmp/psh
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float4v __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, int iters, float seed)
{
    const int lane = threadIdx.x;
    float sh[48], basis[16];
    for (int i = 0; i < 48; i++) sh[i] = seed * 0.001f * (i + 1) + lane * 1e-4f;
    for (int i = 0; i < 16; i++) basis[i] = seed * 0.01f * (i + 1) - lane * 1e-5f;
    float r = 0.f, g = 0.f, b = 0.f;
    float4v acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {
#pragma unroll
            for (int kk = 0; kk < 16; kk++) { r = fmaf(basis[kk], sh[3 * kk], r); g = fmaf(basis[kk], sh[3 * kk + 1], g); b = fmaf(basis[kk], sh[3 * kk + 2], b); }
            asm volatile("" : "+v"(r), "+v"(g), "+v"(b));
        } else if (KIND == 1) {
            // 4 groups of 16 Gaussians; lane supplies A = sh[k][lane % 4] of Gaussian (lane / 4) of the group, B = basis_k
#pragma unroll
            for (int grp = 0; grp < 4; grp++)
#pragma unroll
                for (int kk = 0; kk < 16; kk++)
                    acc[grp] = __builtin_amdgcn_mfma_f32_4x4x1f32(sh[(3 * kk + grp) % 48], basis[kk], acc[grp], 0, 0, 0);
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        } else {
            // 16x16x4: one instruction contracts 4 coefficients for ONE Gaussian's 3 channels (16 Gaussians would need 16
            // different B operands); 64 Gaussians x 4 k-steps = 256 instructions
#pragma unroll
            for (int gg = 0; gg < 16; gg++)       // 16 of the 64 Gaussians per outer iteration, x4 below via iters scaling
#pragma unroll
                for (int ks = 0; ks < 4; ks++)
                    acc[gg & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(sh[(gg + ks) % 48], basis[ks * 4 + (gg & 3)], acc[gg & 3], 0, 0, 0);
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        }
    }
    out[blockIdx.x * 64 + lane] = r + g + b + acc[0].x + acc[1].y + acc[2].z + acc[3].w;
}

template <int KIND>
static void run(const char* name, float* d, double insts_per_64)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    printf("%-34s", name);
    for (int wps : {1, 2, 4}) {
        const int blocks = cus * 4 * wps, iters = 2000;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 10, 1.0f);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0f);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        // one loop iteration = SH -> RGB of 64 Gaussians (KIND 2: of 16 Gaussians -> x4)
        const double per64 = ms * 1e-3 * mhz * 1e6 / ((double)wps * iters) * (KIND == 2 ? 4.0 : 1.0);
        printf("  %dw/SIMD: %7.0f cyc", wps, per64);
    }
    printf("   SIMD cycles per 64 Gaussians (%.0f instructions)\n", insts_per_64);
}

int main()
{
    float* d; (void)hipMalloc(&d, (size_t)16 << 20);
    run<0>("VALU 48 x v_fma_f32", d, 48);
    run<1>("MFMA 64 x v_mfma_f32_4x4x1_16B_f32", d, 64);
    run<2>("MFMA 256 x v_mfma_f32_16x16x4_f32", d, 256);
    return 0;
}
