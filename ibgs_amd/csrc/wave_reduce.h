// wave64 butterfly transpose-reduce (shared by render_bwd.hip and tests/csrc/test_wave_reduce.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace ibgs {

// ---- wave64 butterfly transpose-reduce of 16 values per lane -----------------------------------
#define IBGS_DPP(old, src, ctrl, bank) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (old)), __builtin_bit_cast(int, (src)), (ctrl), 0xF, (bank), false))

// Per-lane select on a wave-uniform lane mask held in an SGPR pair: x in the lanes whose bit is set, 0 elsewhere.
// Written as asm because hipcc routes such masks through VCC and emits the VOP2 form `v_cndmask_b32 v, 0, v, vcc`,
// which issues at ~23 cycles per wave on gfx950 against ~4 for the VOP3 form with an SGPR pair
// (tests/csrc/probe_valu_rate.hip).
__device__ __forceinline__ float select_or_zero(unsigned long long mask, float x)
{
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(mask));
    return r;
}

// The same, for an x that comes straight out of a transcendental (v_exp_f32 ...): hipcc pads no wait states inside inline asm, and a
// non-transcendental reader of a transcendental's result needs one (cdna_hip_programming.md 5.7) -- the s_nop supplies it.
__device__ __forceinline__ float select_or_zero_after_trans(unsigned long long mask, float x)
{
    float r;
    asm("s_nop 0\n\tv_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(mask));
    return r;
}

// min(0.99, x) as one instruction: across a basic-block boundary hipcc puts a canonicalising v_max_f32 x, x in front
// of fminf (MI355X_MICROARCH.md, "canonicalising v_max").  NaN in -> 0.99 out, like fminf.
__device__ __forceinline__ float min_099(float x)
{
    float r;
    asm("v_min_f32 %0, 0x3f7d70a4, %1" : "=v"(r) : "v"(x));
    return r;
}

// v_permlane32_swap: lanes [32,63] of `a` <-> lanes [0,31] of `b`; v_permlane16_swap: odd 16-lane rows of
// `a` <-> even rows of `b` (lane maps verified on hardware by tests/csrc/probe_dpp.hip).  Inline asm because
// hipcc (ROCm 7.2) mis-assigns the second result of __builtin_amdgcn_permlane{16,32}_swap (it emitted
// v_add v2, v2, v2 for a + b).  A VALU write of a swap operand needs 2 wait states before the swap reads it
// (cdna_hip_programming.md T21): all swaps of one butterfly stage sit in ONE asm block behind a single
// s_nop 1 -- they touch disjoint registers, so no swap depends on the one before it.
#define IBGS_SW32(a, b) "v_permlane32_swap_b32 %" #a ", %" #b "\n\t"
#define IBGS_SW16(a, b) "v_permlane16_swap_b32 %" #a ", %" #b "\n\t"

// Lane layout of the results (both reducers): a lane's 16-lane row r = lane >> 4 owns the values
// base(r) .. base(r) + VPR - 1 (VPR values per row); see reduce12_column / reduce16_column.

// ---- 12 values (colour backward: 11 live) ------------------------------------------------------
// In: v[0..11] per lane.  Out: in the lanes where reduce12_column(lane) = c >= 0, the sum over all 64 lanes
// of v[c]; 31 VALU instructions.
__device__ __forceinline__ int reduce12_column(int lane)
{
    const int base = (lane >> 5) * 6 + ((lane >> 4) & 1) * 3, l = lane & 15;
    return l == 0 ? base : (l == 8 ? base + 1 : (l == 1 ? base + 2 : -1));
}
__device__ __forceinline__ float wave_transpose_reduce12(float (&v)[12], int lane)
{
    // lane bit 5 (halves of 32): lower half keeps v[0..5], upper half v[6..11]
    asm volatile("s_nop 1\n\t" IBGS_SW32(0, 6) IBGS_SW32(1, 7) IBGS_SW32(2, 8) IBGS_SW32(3, 9) IBGS_SW32(4, 10) IBGS_SW32(5, 11)
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]),
                   "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]));
#pragma unroll
    for (int i = 0; i < 6; i++) v[i] += v[i + 6];
    // lane bit 4 (rows of 16): even rows keep v[0..2], odd rows v[3..5]
    asm volatile("s_nop 1\n\t" IBGS_SW16(0, 3) IBGS_SW16(1, 4) IBGS_SW16(2, 5)
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
#pragma unroll
    for (int i = 0; i < 3; i++) v[i] += v[i + 3];
    // lane bit 3: fold the two 8-lane halves of each row; lanes 0-7 then carry value 0, lanes 8-15 value 1, all 16 value 2
    const float t0 = v[0] + IBGS_DPP(0.f, v[0], 0x128 /* row_ror:8 */, 0xF);
    const float t1 = v[1] + IBGS_DPP(0.f, v[1], 0x128, 0xF);
    float A = (lane & 8) ? t1 : t0;
    float B = v[2] + IBGS_DPP(0.f, v[2], 0x128, 0xF);
    // lane bits 2..0: 8 lanes -> 1
    A += IBGS_DPP(0.f, A, 0x141 /* row_half_mirror */, 0xF); B += IBGS_DPP(0.f, B, 0x141, 0xF);
    A += IBGS_DPP(0.f, A, 0xB1 /* quad_perm [1,0,3,2] */, 0xF); B += IBGS_DPP(0.f, B, 0xB1, 0xF);
    A += IBGS_DPP(0.f, A, 0x4E /* quad_perm [2,3,0,1] */, 0xF); B += IBGS_DPP(0.f, B, 0x4E, 0xF);
    return (lane & 15) == 1 ? B : A;
}

// ---- 16 values (geo backward: 15 live) ---------------------------------------------------------
// In: v[0..15] per lane.  Out: in the lanes where reduce16_column(lane) = c >= 0, the sum over all 64 lanes of v[c].
__device__ __forceinline__ int reduce16_column(int lane)
{
    const int base = (lane >> 5) * 8 + ((lane >> 4) & 1) * 4, l = lane & 15;
    return l == 0 ? base : (l == 8 ? base + 2 : (l == 1 ? base + 1 : (l == 9 ? base + 3 : -1)));
}
__device__ __forceinline__ float wave_transpose_reduce16(float (&v)[16], int lane)
{
    asm volatile("s_nop 1\n\t" IBGS_SW32(0, 8) IBGS_SW32(1, 9) IBGS_SW32(2, 10) IBGS_SW32(3, 11) IBGS_SW32(4, 12) IBGS_SW32(5, 13) IBGS_SW32(6, 14) IBGS_SW32(7, 15)
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                   "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] += v[i + 8];
    asm volatile("s_nop 1\n\t" IBGS_SW16(0, 4) IBGS_SW16(1, 5) IBGS_SW16(2, 6) IBGS_SW16(3, 7)
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] += v[i + 4];
    // row holds values 0..3: A = (0 | 2) in its (lower | upper) 8 lanes, B = (1 | 3)
    const float t0 = v[0] + IBGS_DPP(0.f, v[0], 0x128, 0xF), t2 = v[2] + IBGS_DPP(0.f, v[2], 0x128, 0xF);
    const float t1 = v[1] + IBGS_DPP(0.f, v[1], 0x128, 0xF), t3 = v[3] + IBGS_DPP(0.f, v[3], 0x128, 0xF);
    float A = (lane & 8) ? t2 : t0, B = (lane & 8) ? t3 : t1;
    A += IBGS_DPP(0.f, A, 0x141, 0xF); B += IBGS_DPP(0.f, B, 0x141, 0xF);
    A += IBGS_DPP(0.f, A, 0xB1, 0xF); B += IBGS_DPP(0.f, B, 0xB1, 0xF);
    A += IBGS_DPP(0.f, A, 0x4E, 0xF); B += IBGS_DPP(0.f, B, 0x4E, 0xF);
    return (lane & 1) ? B : A;
}

}  // namespace ibgs
