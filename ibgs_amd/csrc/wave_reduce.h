// wave64 butterfly transpose-reduce (shared by render_bwd.hip and tests/csrc/test_wave_reduce.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace ibgs {

// ---- wave64 butterfly transpose-reduce of 16 values per lane -----------------------------------
#define IBGS_DPP(old, src, ctrl, bank) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (old)), __builtin_bit_cast(int, (src)), (ctrl), 0xF, (bank), false))

// v_permlane32_swap: lanes [32,63] of `a` <-> lanes [0,31] of `b`; v_permlane16_swap: odd 16-lane rows of
// `a` <-> even rows of `b` (lane maps verified on hardware by tests/csrc/probe_dpp.hip).  Inline asm because
// hipcc (ROCm 7.2) mis-assigns the second result of __builtin_amdgcn_permlane{16,32}_swap (it emitted
// v_add v2, v2, v2 for a + b); the s_nop covers the VALU-write -> permlane-read wait states (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void swap32(float& a, float& b)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16(float& a, float& b)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// In: v[0..15] per lane.  Out (return value): in lane l the sum over all 64 lanes of v[l >> 2].
__device__ __forceinline__ float wave_transpose_reduce16(float (&v)[16], int lane)
{
    // lane bit 5: halves of 32
#pragma unroll
    for (int i = 0; i < 8; i++) { swap32(v[i], v[i + 8]); v[i] += v[i + 8]; }
    // lane bit 4: rows of 16
#pragma unroll
    for (int i = 0; i < 4; i++) { swap16(v[i], v[i + 4]); v[i] += v[i + 4]; }
    // lane bit 3: partner = lane ^ 8 (row rotate by 8); upper 8 lanes of a row keep v[i+2]
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float own = b3 ? v[i + 2] : v[i];
        float other = IBGS_DPP(0.f, v[i], 0x128 /* row_ror:8 */, 0xF);
        other = IBGS_DPP(other, v[i + 2], 0x128, 0xC /* lanes 8..15 of each row */);
        v[i] = own + other;
    }
    // lane bit 2: partner = lane ^ 4: lanes with bit2 = 0 read lane+4 (row_ror:12), the others lane-4 (row_ror:4)
    {
        const float own = b2 ? v[1] : v[0];
        float other = IBGS_DPP(0.f, v[0], 0x12C /* row_ror:12 */, 0xF);
        other = IBGS_DPP(other, v[1], 0x124 /* row_ror:4 */, 0xA /* banks 1 and 3 */);
        v[0] = own + other;
    }
    // lane bits 1, 0: the four lanes of a quad hold partial sums of the same value
    v[0] += IBGS_DPP(0.f, v[0], 0xB1 /* quad_perm [1,0,3,2] */, 0xF);
    v[0] += IBGS_DPP(0.f, v[0], 0x4E /* quad_perm [2,3,0,1] */, 0xF);
    return v[0];
}


}  // namespace ibgs
