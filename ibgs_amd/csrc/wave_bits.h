// wave64 transpose of a 64 x 64 bit matrix held as one row per lane (used by binning.hip and tests/csrc/test_wave_bits.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ibgs {

// Per-lane constants of the transpose (hoisted out of loops by the caller): for the stages that exchange with lane ^ j,
// j = 8, 4, 2, 1: which bits a lane keeps and by how much it rotates what it receives.
struct BitTransposeConsts { uint32_t keep[4], rot[4]; };

__device__ __forceinline__ BitTransposeConsts bit_transpose_consts(int lane)
{
    BitTransposeConsts c;
    const uint32_t M[4] = {0x00FF00FFu, 0x0F0F0F0Fu, 0x33333333u, 0x55555555u};
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int j = 8 >> s;
        c.keep[s] = (lane & j) ? ~M[s] : M[s];
        c.rot[s] = (lane & j) ? (uint32_t)j : (uint32_t)(32 - j);        // rotate right: by j in the upper lane of a pair, by 32 - j (= left by j) in the lower
    }
    return c;
}

#define IBGS_DPPU(old, src, ctrl, bank) \
    ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(src), (ctrl), 0xF, (bank), false))

// In: lane l holds row l (bit t of {hi, lo} = element (l, t)).  Out: lane t holds column t (bit l = element (l, t)).
// Six butterfly stages (lane ^ 32, 16, 8, 4, 2, 1), 31 VALU instructions:
//   32: one v_permlane32_swap (the upper word of the lower lane <-> the lower word of the upper lane);
//   16: a swap gathers both lower words in the even row and both upper words in the odd row, two byte permutes form both results,
//       a second swap hands them back;
//   8..1: the partner's word by DPP, rotated, merged with v_bfi.
__device__ __forceinline__ void wave_bit_transpose64(uint32_t& lo, uint32_t& hi, const BitTransposeConsts& c)
{
    // (inline asm: see wave_reduce.h -- hipcc mis-assigns the second result of the swap builtins; a VALU write of an operand needs
    // two wait states before the swap reads it)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
    {
        // even row: (lo, hi) = (own lower word, partner's lower word); odd row: (partner's upper word, own upper word)
        const uint32_t x = __builtin_amdgcn_perm(hi, lo, 0x05040100u);      // halves [lo.l, hi.l]: result of the even lane
        const uint32_t y = __builtin_amdgcn_perm(hi, lo, 0x07060302u);      // halves [lo.h, hi.h]: result of the odd lane
        lo = x; hi = y;
    }
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
#define IBGS_BT_STAGE(s, fetch)                                                                      \
    {                                                                                                \
        uint32_t pl, ph;                                                                             \
        fetch;                                                                                       \
        lo = (c.keep[s] & lo) | (~c.keep[s] & __builtin_amdgcn_alignbit(pl, pl, c.rot[s]));           \
        hi = (c.keep[s] & hi) | (~c.keep[s] & __builtin_amdgcn_alignbit(ph, ph, c.rot[s]));           \
    }
    IBGS_BT_STAGE(0, (pl = IBGS_DPPU(0, lo, 0x128 /* row_ror:8 */, 0xF), ph = IBGS_DPPU(0, hi, 0x128, 0xF)))
    IBGS_BT_STAGE(1, (pl = IBGS_DPPU(IBGS_DPPU(0, lo, 0x104 /* row_shl:4 */, 0x5), lo, 0x114 /* row_shr:4 */, 0xA),
                      ph = IBGS_DPPU(IBGS_DPPU(0, hi, 0x104, 0x5), hi, 0x114, 0xA)))
    IBGS_BT_STAGE(2, (pl = IBGS_DPPU(0, lo, 0x4E /* quad_perm [2,3,0,1] */, 0xF), ph = IBGS_DPPU(0, hi, 0x4E, 0xF)))
    IBGS_BT_STAGE(3, (pl = IBGS_DPPU(0, lo, 0xB1 /* quad_perm [1,0,3,2] */, 0xF), ph = IBGS_DPPU(0, hi, 0xB1, 0xF)))
#undef IBGS_BT_STAGE
}

}  // namespace ibgs
