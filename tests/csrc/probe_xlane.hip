// Cross-lane instruction cost probe for gfx950 (cycles per wave64 instruction per SIMD at 1..8 waves per SIMD):
// what the butterfly transpose-reduce of render_bwd.hip is made of.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/px tests/csrc/probe_xlane.hip && /tmp/px
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {          // baseline: v_add_f32
            REP8(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 1) {   // v_permlane32_swap on 4 disjoint pairs, twice (8 instructions)
            REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 2) {   // v_permlane16_swap
            REP8(asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                              "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 3) {   // v_add_f32_dpp row_ror:8
            REP8(asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 4) {   // v_add_f32_dpp row_bcast:15 (wave-level broadcast form)
            REP8(asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %4, %4 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 5) {   // s_nop 0 between adds: 8 x (v_add, s_nop 0) counted as 8 "instructions" (pairs)
            REP8(asm volatile("v_add_f32 %0, %0, %1\n s_nop 0\n v_add_f32 %1, %1, %2\n s_nop 0\n v_add_f32 %2, %2, %3\n s_nop 0\n v_add_f32 %3, %3, %4\n s_nop 0\n v_add_f32 %4, %4, %5\n s_nop 0\n v_add_f32 %5, %5, %6\n s_nop 0\n v_add_f32 %6, %6, %7\n s_nop 0\n v_add_f32 %7, %7, %0\n s_nop 0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 6) {   // s_nop 1 pairs
            REP8(asm volatile("v_add_f32 %0, %0, %1\n s_nop 1\n v_add_f32 %1, %1, %2\n s_nop 1\n v_add_f32 %2, %2, %3\n s_nop 1\n v_add_f32 %3, %3, %4\n s_nop 1\n v_add_f32 %4, %4, %5\n s_nop 1\n v_add_f32 %5, %5, %6\n s_nop 1\n v_add_f32 %6, %6, %7\n s_nop 1\n v_add_f32 %7, %7, %0\n s_nop 1\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 7) {   // ds_swizzle (LDS crossbar, no memory): swap halves of 32 via BitMode xor 0x10
            REP8(asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM,\"0000p\")\n ds_swizzle_b32 %1, %1 offset:swizzle(BITMASK_PERM,\"0000p\")\n ds_swizzle_b32 %2, %2 offset:swizzle(BITMASK_PERM,\"0000p\")\n ds_swizzle_b32 %3, %3 offset:swizzle(BITMASK_PERM,\"0000p\")\n"
                              "ds_swizzle_b32 %4, %4 offset:swizzle(BITMASK_PERM,\"0000p\")\n ds_swizzle_b32 %5, %5 offset:swizzle(BITMASK_PERM,\"0000p\")\n ds_swizzle_b32 %6, %6 offset:swizzle(BITMASK_PERM,\"0000p\")\n ds_swizzle_b32 %7, %7 offset:swizzle(BITMASK_PERM,\"0000p\")\n s_waitcnt lgkmcnt(0)\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 8) {   // v_mov_b32_dpp quad_perm
            REP8(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 9) {   // global_atomic_add_f32 from 11 lanes to one 64-byte row (the kernel's shape), no waits
            if (threadIdx.x < 11) { REP8(atomicAdd(out + 1024 + ((blockIdx.x * 8 + (i & 7)) & 65535) * 16 + threadIdx.x, a0);) }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
static void run(const char* name, float* d)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    printf("%-28s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * 4 * wps, iters = KIND == 9 ? 200 : 2000;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 10, 1.0f);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0f);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)wps * iters * (KIND == 9 ? 8 : 64);
        printf("  %dw: %.2f", wps, ms * 1e-3 * mhz * 1e6 / instr_per_simd);
    }
    printf("   cycles per wave64 instruction per SIMD (nominal %.0f MHz)\n", mhz);
}

int main()
{
    float* d; (void)hipMalloc(&d, (size_t)64 << 20);
    (void)hipMemset(d, 0, (size_t)64 << 20);
    run<0>("v_add_f32", d); run<1>("v_permlane32_swap_b32", d); run<2>("v_permlane16_swap_b32", d); run<3>("v_add_f32_dpp row_ror:8", d);
    run<4>("v_add_f32_dpp row_bcast", d); run<5>("v_add_f32 + s_nop 0", d); run<6>("v_add_f32 + s_nop 1", d); run<7>("ds_swizzle_b32", d);
    run<8>("v_mov_b32_dpp quad_perm", d); run<9>("atomic 11 lanes / 64 B row", d);
    return 0;
}
