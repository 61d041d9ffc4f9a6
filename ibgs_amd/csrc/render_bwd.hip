// B1 / B2: back-to-front re-traversal of every tile list, gradients w.r.t. the per-Gaussian render
// record (2D mean, conic, opacity, colour, plane parameters).
//
// Behaviour: DPR/cuda_rasterizer/backward.cu:496-807 (renderCUDA) + bilinearInterpolateBackward
// (backward.cu:55-109).  The reference issues up to 16 global atomicAdd per (pixel, Gaussian) pair
// into five separate arrays.  On MI355X global float atomics run at ~1.3 TB/s chip-wide only when
// they arrive as contiguous 64-byte requests (MI355X_MICROARCH.md "Global float atomics"), and the blend
// kernels are VALU-issue bound, so here:
//
//   * one wave owns a 16x16 tile (4 pixels per lane: lane l = pixel (l%8, l/8) of each 8x8 quadrant) or, on frames with
//     fewer than 4096 tiles, one 8x8 quadrant (the chip would stay mostly empty otherwise); no cross-wave synchronisation;
//   * per pixel and Gaussian a lane keeps ONE number, Q = o G dL/dalpha; the six geometric moments sum Q, Q d, Q d d^T
//     (d = Gaussian centre - pixel) are assembled once per Gaussian from the lane's four Q and its d0 (the other pixels sit at
//     d0 - (8,0), (0,8), (8,8)); the per-Gaussian constants (conic, opacity, 0.5 W) are applied in preprocess_bwd.hip.
//     Same mathematics as the reference's eight per-pair quantities (the sums are linear), ~24 instead of ~41 VALU
//     instructions per pair;
//   * the 12 (colour) / 16 (geo) per-lane partial sums are reduced over the 64 lanes with a butterfly TRANSPOSE-reduce
//     (v_permlane32_swap, v_permlane16_swap, DPP row rotates / mirrors, quad_perm; wave_reduce.h), after which one lane per
//     value holds its wave total -- no LDS traffic; measured cost ~145 cycles per Gaussian (profiles/r02_probe_xlane.txt);
//   * the wave then issues ONE atomic instruction per Gaussian: 11 (colour) / 15 (geo) lanes add their totals to that
//     Gaussian's 64-byte accumulation row (grad_acc[P][16]) -- one 64-byte memory-side request per (Gaussian, tile)
//     instead of 11-16 scattered dword atomics per (Gaussian, pixel).  IBGS_FLAG_DETERMINISTIC stores the totals in a slab
//     instead (deterministic.hip);
//   * Gaussians that no pixel of the wave uses (ballot == 0, or behind every pixel's last contributor) cost a few VALU
//     instructions and no memory traffic;
//   * geo: the median / warp block (B2) runs as a pixel-parallel pre-pass that leaves a per-pixel table; the blend loop only
//     looks entries up (geo_window_kernel below).
//
// Deviations (documented in DESIGN.md): the skip test is the forward's single compare E <= log2(255) on the same staged numbers
// (render_fwd.hip, common.h), so both passes visit exactly the same (pixel, Gaussian) pairs; alpha is recomputed with the
// same fast exp2 as the forward (the reference uses __expf forward / exp backward); 1/(1-alpha) is the hardware
// reciprocal (1 ulp) instead of an IEEE division.
#include "common.h"
#include <stdlib.h>
#include "wave_reduce.h"
#include <type_traits>

namespace ibgs {

struct BwdParams {
    const uint32_t* ranges; const uint32_t* point_list; const float4* rec;
    Cam cam;
    int ntiles;
    TileMap tmap;
    int n_src; int tex_quant;
    int power_skip;       // RA_* bits (below): 0 = no reference branch at all (IBGS_FLAG_NO_REF_POWER_SKIP), else how the pairs of near-singular conics are evaluated
    const uint32_t* tile_risky;      // 4 words per tile, written by the forward (ImgState::tile_risky): non-zero = the tile's list holds a near-singular conic.  Those tiles belong to
                                     // the *_risk_kernel launched beside every blend kernel; the fast kernels skip them (nullptr: nobody skips anything)
    const float* ref_to_src; const float4* src_rgba;
    const float* final_T; const uint32_t* n_contrib; const float* sum_w; const uint32_t* low_high;
    const int32_t* valid_idx; const float* valid_w;
    const float* depth_pixels; const float* warped_pixels;
    const float* dL_dcolor; const float* dL_dnormal; const float* dL_ddepth; const float* dL_dwarped;
    float* gacc;
    int tab_slots;        // geo: slots of the window table the caller's scratch holds (buffer_length + 1 when the caller stated it, else 8)
    const uint32_t* slot_c; const uint32_t* meta; float* tab;     // geo: the forward's buffered contributor numbers (+ slot count); the window pass's table
    const uint32_t* order;      // balanced launch order of the colour kernel: workgroup -> tile (0xFFFFFFFF: none), nullptr: the tile map decides
    int slab_ipt;         // rows per list entry in `slab` when that is not the body's own waves per tile (hybrid kernel: 4), else 0
    int hybrid_grid1;     // hybrid colour kernel: slots of `order` = workgroups that are a tile's first wave (hybrid_item, common.h)
    float* slab;          // IBGS_FLAG_DETERMINISTIC: (R x waves per tile) x 16, one row per (list entry, wave of its tile), written instead of the atomics (else nullptr)
};

__device__ __forceinline__ float quant8b(float a, int quant) { return quant ? floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f) : a; }

__device__ __forceinline__ float4 tex_rgba_b(const float4* __restrict__ img, int W, int H, float x, float y, int quant)
{
#pragma clang fp contract(off)          // operation by operation like the oracle's tex_rgb (the pass that calls it is bound by its gathers)
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fxi = floorf(xb), fyi = floorf(yb);
    const float a = quant8b(xb - fxi, quant), b = quant8b(yb - fyi, quant);
    const int i0 = min(W - 1, max(0, (int)fxi)), i1 = min(W - 1, max(0, (int)fxi + 1));
    const int j0 = min(H - 1, max(0, (int)fyi)), j1 = min(H - 1, max(0, (int)fyi + 1));
    const float4 t00 = img[(size_t)j0 * W + i0], t10 = img[(size_t)j0 * W + i1];
    const float4 t01 = img[(size_t)j1 * W + i0], t11 = img[(size_t)j1 * W + i1];
    const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
    float4 r;
    r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
    r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
    r.z = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
    r.w = 1.0f;
    return r;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, WAVE));
    return v;
}


__device__ __forceinline__ float fast_rcp(float x)
{   // v_rcp_f32 (1 ulp) + one Newton step
    const float r = __builtin_amdgcn_rcpf(x);
    return r * (2.0f - x * r);
}


// ---- near-singular conics (round 6) --------------------------------------------------------------------------------------------------------
// A conic within 10^-3 of singular (common.h: conic_takes_ref_power) is where fp32 shows in the gradients: the chain of backward.cu:405-420 forms
// dL/dcov2D = (-c^2 Sxx + 2bc Sxy - b^2 Syy) / det^2 from the pixel sums S = sum q d d^T, a difference of near-equal numbers that turns every ulp
// of S into 10^3-10^4 ulp of the result.  Measured on the needle scenes of tests/test_gpu_fuzz_pins.py (profiles/r06_ref_arith_ab.txt): ANY
// fp32 evaluation of that chain -- the oracle's two builds, its float-sum build, the kernels with libm exp / IEEE division / the reference's association
// in any combination, with one summation order or another -- lands 1e-3 .. 3e-2 from the float64 evaluation, and which one is closest changes with the
// order of the additions alone (ratio to the fp32 oracle builds: 0.8 .. 3.9 on one scene).  No per-pair value carries that; the chain does.  So for those
// Gaussians the blend sums what the chain really needs,
//        dL/dcov2D = 0.5 sum_pairs q l l^T,   l = conic d   (algebraically the same: -c^2 dx^2 + 2bc dx dy - b^2 dy^2 = -(c dx - b dy)^2 = -(det l_x)^2),
// per pixel, where nothing cancels (RA_LFORM, the default: l_moments below, a wave-uniform branch of the fast kernels; preprocess_bwd.hip takes the three
// sums as dL/dcov2D): 3-10 x closer to float64 than the reference's own fp32 arithmetic and the same from run to run, whatever the order.
// IBGS_FLAG_REF_ARITH selects the reference's arithmetic to the letter instead (RA_ASSOC; SURVEY Q1 as a switch): G = exp(power) at libm accuracy
// (backward.cu:648), T / (1 - alpha) as an IEEE division (:654), the eight per-pair quantities of :779-804 in its association, uncontracted; the row then
// holds the reference's sums and the ill-conditioned chain runs as the reference runs it.  That block (libm exp, divisions, eight more sums) does not fit
// beside the fast path -- inlined it cost the hot loops 19-40 spilled VGPRs, as a called function 60 % of the kernel's time -- so under the flag the tiles
// whose list holds such a conic (flagged by the forward: ImgState::tile_risky) are walked by kernels of their own (*_risk_kernel: the RISK = true
// instantiation of the bodies below, launched behind the blend kernel, which skips those tiles).
// The DECISIONS (which pairs blend) are the forward's either way: both passes visit the same pairs.
constexpr int RA_DECIDE = 1, RA_LFORM = 2, RA_ASSOC = 4;
__device__ __forceinline__ float ref_power(float dx, float dy, float a, float b, float c)
{
#pragma clang fp contract(off)
    return -0.5f * (a * dx * dx + c * dy * dy) - b * dx * dy;          // forward.cu:419, backward.cu:644
}
// RA_ASSOC: rs[0..1] dL/dG dG/ddel{x,y} (x 0.5 W / 0.5 H in preprocess_bwd), rs[2..3] their magnitudes, rs[4..6] gd{x,x,y} d{x,y,y} dL/dG (x -0.5 there), rs[7] G dL/dalpha
__device__ __forceinline__ void ref_pair_sums(float (&rs)[8], float G, float dL_dalpha, float dx, float dy, float o, float a, float b, float c)
{
#pragma clang fp contract(off)
    const float dL_dG = o * dL_dalpha;
    const float gdx = G * dx, gdy = G * dy;
    const float dG_ddelx = -gdx * a - gdy * b;
    const float dG_ddely = -gdy * c - gdx * b;
    const float mx = dL_dG * dG_ddelx, my = dL_dG * dG_ddely;
    rs[0] += mx; rs[1] += my; rs[2] += fabsf(mx); rs[3] += fabsf(my);
    rs[4] += gdx * dx * dL_dG; rs[5] += gdx * dy * dL_dG; rs[6] += gdy * dy * dL_dG;
    rs[7] += G * dL_dalpha;
}
// The six geometric sums of a near-singular conic on the fast path, in l-space: v[0..1] = sum q l, v[4..6] = sum q l l^T, v[7] = sum q over the lane's pixels, with
// l = (scaled conic) d formed per pixel -- the quadrant-0 value shifted like E (d_q = d_0 - (8,0), (0,8), (8,8)).  What cancels catastrophically downstream of the
// d-moments (K S K with K = conic) never enters: l is small where the Gaussian is.  ~30 instructions instead of 17, for these Gaussians only (a wave-uniform branch).
// preprocess_bwd.hip undoes the exp2 scale of the conic (EXP2_UNSCALE, squared for the second moments) and takes the second moments as dL/dcov2D.
template <int PPL, int NV>
__device__ __forceinline__ void l_moments(float (&v)[NV], const float (&Q)[PPL], float dx0, float dy0, float ca, float cb, float cc)
{
    const float lx0 = ca * dx0 + cb * dy0, ly0 = cb * dx0 + cc * dy0;
    float sx = 0.f, sy = 0.f, sxx = 0.f, sxy = 0.f, syy = 0.f, s0 = 0.f;
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        // pixel offsets of the lane's quadrants: PPL == 4: (0,0), (8,0), (0,8), (8,8); PPL == 2: (0,0), (8,0)
        const float ox = (PPL == 4) ? (float)((q & 1) * 8) : (float)(q * 8), oy = (PPL == 4) ? (float)((q >> 1) * 8) : 0.f;
        const float lx = fmaf(-ox, ca, fmaf(-oy, cb, lx0)), ly = fmaf(-ox, cb, fmaf(-oy, cc, ly0));
        const float qx = Q[q] * lx, qy = Q[q] * ly;
        sx += qx; sy += qy; s0 += Q[q];
        sxx = fmaf(qx, lx, sxx); sxy = fmaf(qx, ly, sxy); syy = fmaf(qy, ly, syy);
    }
    v[0] = sx; v[1] = sy; v[4] = sxx; v[5] = sxy; v[6] = syy; v[7] = s0;
}
// is `tile` one of the flagged tiles?  (four words per tile: one per wave of the forward variant that walked it, unused ones zero)
__device__ __forceinline__ bool tile_is_risky(const uint32_t* __restrict__ tile_risky, int tile)
{
    if (!tile_risky) return false;
    const uint4 w = *reinterpret_cast<const uint4*>(tile_risky + (size_t)tile * 4);
    return (w.x | w.y | w.z | w.w) != 0u;
}

// ---- colour variant -------------------------------------------------------------------------------------------------
// Same traversal and the same (pixel, Gaussian) decisions as render_bwd_body below, restructured so that a pair costs
// ~24 VALU instructions instead of ~41 (the kernel is VALU-issue bound):
//   * what a lane keeps per quadrant is ONE number, Q_q = o G dL/dalpha of its pixel there (0 when the quadrant is skipped);
//     the six geometric moments are formed once per Gaussian from Q_0..Q_3 and the lane's d0 = (Gaussian centre - its
//     quadrant-0 pixel): the other three pixels sit at d0 - (8,0), (0,8), (8,8), so
//         sum Q d   = d0 S0 - 8 (A, B),                      A = Q1 + Q3, B = Q2 + Q3, S0 = Q0 + Q1 + Q2 + Q3
//         sum Q dx^2 = dx0 (dx0 S0 - 16 A) + 64 A            (likewise y)
//         sum Q dx dy = dy0 Sx - 8 (dx0 B - 8 Q3)
//     -- 17 instructions per Gaussian instead of 8 per (Gaussian, quadrant);
//   * conic * d of the other quadrants (only the |.| moments need it) comes from the quadrant-0 value by one fma each;
//   * the per-pixel constants dL/dC and -T_final (bg . dL/dC) travel as ONE 16-byte LDS read;
//   * one select (on G) instead of two, 1/(1 - alpha) is the bare v_rcp_f32 (1 ulp; the Newton step bought nothing that the
//     parity bars see: T picks up ~1e-7 sqrt(k) relative noise over k divisions);
//   * "k < n_contrib" costs two compares per quadrant per CHUNK when no pixel of the quadrant switches on inside the chunk
//     (each pixel switches on once per traversal), instead of one per Gaussian.
#ifdef IBGS_COUNT_LANES
__device__ unsigned long long g_lanes_bwd[4];
#endif
constexpr int BWD_CHUNK = 16;          // 16 records per round: 0.75 KB + 4 KB of per-pixel constants <= 5 KB per wave = 8 waves per SIMD, every tile of a 1080p frame resident at once
template <int PPL, bool ABS = true, bool RISK = false>          // ABS = false (IBGS_FLAG_NO_ABS_GRAD): the |.| moments of dL/dmean2D are not accumulated; RISK: the tile's list holds near-singular conics (see above)
__device__ __forceinline__ void render_bwd_color_body(const BwdParams& p, const int tile, const int sub, float4 (&s_rec)[3][BWD_CHUNK],
                                                      float4 (*s_gpix)[WAVE] /* PPL rows: dL/dC (rgb), -T_final * (bg . dL/dC) */)
{
    // (the LDS is the caller's: the hybrid kernel below holds two instantiations of this body and hands both the same arrays)
    IBGS_LANES_DECL();
    constexpr int CHUNK = BWD_CHUNK;

    const int lane = threadIdx.x;
    int col = reduce12_column(lane);
    if (col >= 11) col = -1;
    constexpr int IPT = 4 / PPL;
    const int quad0 = sub * PPL;
    const int W = p.cam.W, H = p.cam.H;
    const int tx0 = (tile % p.cam.gx) * TILE, ty0 = (tile / p.cam.gx) * TILE;
    const size_t HW = (size_t)W * H;

    float T[PPL], S[PPL];
    const bool bg0 = p.cam.bg[0] == 0.f && p.cam.bg[1] == 0.f && p.cam.bg[2] == 0.f;      // wave-uniform (scalar loads)
    uint32_t ncontrib[PPL];
    uint32_t nmax = 0;
    const float pxf0 = (float)(tx0 + (quad0 & 1) * 8 + (lane & 7)), pyf0 = (float)(ty0 + (quad0 >> 1) * 8 + (lane >> 3));
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        const int qq = quad0 + q;
        const int px = tx0 + (qq & 1) * 8 + (lane & 7), py = ty0 + (qq >> 1) * 8 + (lane >> 3);
        const bool inside = px < W && py < H;
        const size_t pix = (size_t)py * W + px;
        const float T_final = inside ? p.final_T[pix] : 0.f;
        T[q] = T_final;
        ncontrib[q] = inside ? p.n_contrib[pix] : 0u;
        nmax = max(nmax, ncontrib[q]);
        S[q] = 0.f;
        float4 g;
        g.x = (inside && p.dL_dcolor) ? p.dL_dcolor[pix] : 0.f;
        g.y = (inside && p.dL_dcolor) ? p.dL_dcolor[HW + pix] : 0.f;
        g.z = (inside && p.dL_dcolor) ? p.dL_dcolor[2 * HW + pix] : 0.f;
        g.w = -T_final * (p.cam.bg[0] * g.x + p.cam.bg[1] * g.y + p.cam.bg[2] * g.z);
        s_gpix[q][lane] = g;
    }
    nmax = wave_max_u32(nmax);
    const uint32_t r0 = p.ranges[2 * tile], r1 = p.ranges[2 * tile + 1];
    const int n = (int)(r1 - r0);
    // entries >= top contribute to no pixel of this wave; readfirstlane makes the loop control scalar (nmax is the same in
    // every lane after the butterfly, which the compiler cannot see)
    int top = __builtin_amdgcn_readfirstlane(min((int)nmax, n));

    while (top > 0) {
        const int count = min(CHUNK, top);
        bool risky = false;
        if (lane < count) {   // stage in processing order: slot l holds entry top-1-l
            const uint32_t id = p.point_list[r0 + (uint32_t)(top - 1 - lane)];
            const float4* r = p.rec + (size_t)id * 4;
            float4 ra = r[0];
            ra.w = __uint_as_float(id);            // the record's spare slot carries the Gaussian index to the atomic
            float4 c1 = r[1];
            risky = conic_takes_ref_power(c1.x, c1.y, c1.z);
            stage_for_exp2(ra, c1);                            // as the forward stages them (common.h): same numbers, same decisions
            s_rec[0][lane] = ra; s_rec[1][lane] = c1; s_rec[2][lane] = r[2];
        }
        const uint64_t riskm = p.power_skip ? __builtin_amdgcn_ballot_w64(risky) : 0ull;      // near-singular conics: the forward's rare branch
        // pixels that take part: k < n_contrib.  k runs from top-1 down to top-count in this chunk and a pixel only ever
        // switches ON (at k = n_contrib - 1), so when the masks at both ends agree they hold for the whole chunk.
        uint64_t ncm[PPL];
        bool stable = true;
#pragma unroll
        for (int q = 0; q < PPL; q++) {
            ncm[q] = __builtin_amdgcn_ballot_w64((uint32_t)(top - count) < ncontrib[q]);
            stable = stable && (ncm[q] == __builtin_amdgcn_ballot_w64((uint32_t)(top - 1) < ncontrib[q]));
        }
        __syncthreads();
        auto chunk = [&](auto stable_tag, auto bg0_tag) {
            constexpr bool STABLE = decltype(stable_tag)::value;
            constexpr bool BG0 = decltype(bg0_tag)::value;          // black background: the -T_final (bg . g) / (1 - alpha) term is zero
            for (int j = 0; j < count; j++) {
                const uint32_t k = (uint32_t)(top - 1 - j);          // 0-based position in the tile list
                const float4 q0 = s_rec[0][j], q1 = s_rec[1][j], q2 = s_rec[2][j];
                const float ca = q1.x, cb = q1.y, cc = q1.z, nlo = q0.z;          // scaled conic, -log2(opacity)
                // same evaluation of p2 = d^T conic d as the forward (render_fwd.hip): once per lane, shifted to the
                // other three quadrants, so both passes take identical alpha decisions
                const float dx0 = q0.x - pxf0, dy0 = q0.y - pyf0;
                const float lx0 = ca * dx0 + cb * dy0, ly0 = cb * dx0 + cc * dy0;
                const float P0 = fmaf(dx0, lx0, fmaf(dy0, ly0, nlo));          // the forward's E (render_fwd.hip)
                float p2q[PPL], lxq[PPL], lyq[PPL];
                p2q[0] = P0; lxq[0] = lx0; lyq[0] = ly0;
                if constexpr (PPL >= 2) {
                    p2q[1] = fmaf(-16.0f, lx0, P0 + 64.0f * ca);
                    if constexpr (ABS) { lxq[1] = fmaf(-8.0f, ca, lx0); lyq[1] = fmaf(-8.0f, cb, ly0); }
                }
                if constexpr (PPL == 4) {
                    p2q[2] = fmaf(-16.0f, ly0, P0 + 64.0f * cc);
                    p2q[3] = fmaf(128.0f, cb, p2q[2] + (p2q[1] - P0));          // E3 = E2 + (E1 - E0) + 128 b: three instructions, as the forward
                    if constexpr (ABS) {
                        lxq[2] = fmaf(-8.0f, cb, lx0); lyq[2] = fmaf(-8.0f, cc, ly0);
                        lxq[3] = fmaf(-8.0f, cb, lxq[1]); lyq[3] = fmaf(-8.0f, cc, lyq[1]);
                    }
                }
                if constexpr (!RISK) {
                    if ((riskm >> j) & 1ull) {          // wave-uniform and rare: E from the reference's expression, `power > 0` pairs dropped (common.h); the sums: l_moments below
                        const uint32_t gid = __builtin_amdgcn_readfirstlane(__float_as_uint(q0.w));
                        const float4 g0 = p.rec[(size_t)gid * 4], g1 = p.rec[(size_t)gid * 4 + 1];
#pragma unroll
                        for (int q = 0; q < PPL; q++) {
                            const float ox = (PPL == 4) ? (float)((q & 1) * 8) : (float)(q * 8), oy = (PPL == 4) ? (float)((q >> 1) * 8) : 0.f;
                            p2q[q] = ref_power_E(g0.x - (pxf0 + ox), g0.y - (pyf0 + oy), g1.x, g1.y, g1.z, nlo);
                        }
                    }
                }
                if constexpr (RISK) {
                    if ((riskm >> j) & 1ull) {          // wave-uniform: a near-singular conic in the reference's arithmetic (IBGS_FLAG_REF_ARITH).  E (the decision) from the reference's
                                                        // expression, `power > 0` pairs dropped (common.h); G, alpha, T and the sums as described above
                        const uint32_t gid = __builtin_amdgcn_readfirstlane(__float_as_uint(q0.w));
                        const float4 g0 = p.rec[(size_t)gid * 4], g1 = p.rec[(size_t)gid * 4 + 1];          // the record as preprocess wrote it (unscaled conic, opacity)
                        float rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rR = 0.f, rG = 0.f, rB = 0.f;
                        bool rany = false;
#pragma unroll
                        for (int q = 0; q < PPL; q++) {
                            const float ox = (PPL == 4) ? (float)((q & 1) * 8) : (float)(q * 8), oy = (PPL == 4) ? (float)((q >> 1) * 8) : 0.f;
                            const float dx = g0.x - (pxf0 + ox), dy = g0.y - (pyf0 + oy);
                            const float power = ref_power(dx, dy, g1.x, g1.y, g1.z);
                            const float E = ref_power_E(dx, dy, g1.x, g1.y, g1.z, nlo);          // what both passes decide on (+inf: `power > 0`)
                            const uint64_t okm = __builtin_amdgcn_ballot_w64(k < ncontrib[q]) & __builtin_amdgcn_ballot_w64(E <= ALPHA_SKIP_E);
                            if (okm == 0ull) continue;
                            rany = true;
                            const float Gr = __builtin_amdgcn_inverse_ballot_w64(okm) ? expf(power) : 0.f;          // backward.cu:648; 0 where the test fails: alpha = 0 leaves T and S unchanged
                            const float oG = g0.z * Gr;
                            const float alpha = fminf(0.99f, oG);
                            const float om = 1.f - alpha;
                            T[q] = T[q] / om;          // backward.cu:654
                            const float w = alpha * T[q];
                            const float4 gp = s_gpix[q][lane];
                            const float cg = q2.x * gp.x + q2.y * gp.y + q2.z * gp.z;
                            float dL_dalpha = cg - S[q];
                            S[q] = fmaf(alpha, dL_dalpha, S[q]);
                            rR = fmaf(w, gp.x, rR); rG = fmaf(w, gp.y, rG); rB = fmaf(w, gp.z, rB);
                            dL_dalpha = dL_dalpha * T[q];
                            if constexpr (!BG0) dL_dalpha += gp.w / om;          // ... - T_final (bg . g) / (1 - alpha)
                            ref_pair_sums(rs, Gr, dL_dalpha, dx, dy, g0.z, g1.x, g1.y, g1.z);
                        }
                        if (__builtin_amdgcn_ballot_w64(rany) != 0ull) {
                            float v[12] = {rs[0], rs[1], rs[2], rs[3], rs[4], rs[5], rs[6], rs[7], rR, rG, rB, 0.f};
                            const float tot = wave_transpose_reduce12(v, lane);
                            if (col >= 0) {
                                if (p.slab) p.slab[((size_t)(r0 + k) * (p.slab_ipt ? p.slab_ipt : IPT) + (size_t)sub) * GACC_FLOATS + col] = tot;
                                else atomicAdd(p.gacc + (size_t)gid * GACC_FLOATS + col, tot);
                            }
                        }
                        continue;
                    }
                }
                float Q[PPL], aX = 0.f, aY = 0.f, vR = 0.f, vG = 0.f, vB = 0.f;
                bool any = false;
                IBGS_LANES_ADD(0, 1);
#pragma unroll
                for (int q = 0; q < PPL; q++) {
                    // the forward's test: E <= log2(255) (render_fwd.hip, common.h)
                    const uint64_t live = STABLE ? ncm[q] : __builtin_amdgcn_ballot_w64(k < ncontrib[q]);
                    const uint64_t okm = live & __builtin_amdgcn_ballot_w64(p2q[q] <= ALPHA_SKIP_E);
                    Q[q] = 0.f;
                    if (okm != 0ull) {
                        any = true;
                        IBGS_LANES_ADD(2, 64); IBGS_LANES_ADD(3, __builtin_popcountll(okm));
                        // lanes that fail the test run the same instructions with G = 0: alpha = 0 leaves T and S
                        // unchanged (1 / (1 - 0) = 1 exactly) and every sum receives a zero
                        // (exp2(-E) = o G.  The select is inline asm and the first reader of the transcendental's result: it carries its own
                        // wait state -- hipcc pads none inside asm, cdna_hip_programming.md 5.7)
                        const float oG = select_or_zero_after_trans(okm, __builtin_amdgcn_exp2f(-p2q[q]));
                        const float alpha = min_099(oG);
                        const float rinv = __builtin_amdgcn_rcpf(1.f - alpha);
                        T[q] = T[q] * rinv;
                        const float w = alpha * T[q];
                        const float4 gp = s_gpix[q][lane];
                        // S = (colour behind this Gaussian) . (pixel gradient): scalar form of the reference's per-channel
                        // accum_rec / last_color / last_alpha recurrence (backward.cu:665-669), folded into one fma:
                        // behind_k = alpha_k c_k + (1 - alpha_k) behind_{k+1} = behind_{k+1} + alpha_k (c_k - behind_{k+1})
                        const float cg = q2.x * gp.x + q2.y * gp.y + q2.z * gp.z;
                        float dL_dalpha = cg - S[q];
                        S[q] = fmaf(alpha, dL_dalpha, S[q]);
                        vR = fmaf(w, gp.x, vR); vG = fmaf(w, gp.y, vG); vB = fmaf(w, gp.z, vB);
                        if constexpr (BG0) dL_dalpha = dL_dalpha * T[q];
                        else dL_dalpha = fmaf(dL_dalpha, T[q], gp.w * rinv);      // ... - T_final (bg . g) / (1 - alpha)
                        const float qv = oG * dL_dalpha;                          // dL/dG * G
                        Q[q] = qv;
                        if constexpr (ABS) { aX = fmaf(fabsf(qv), fabsf(lxq[q]), aX); aY = fmaf(fabsf(qv), fabsf(lyq[q]), aY); }      // |a b| = |a| |b|: one v_fma with source modifiers
                    }
                }
                if (__builtin_amdgcn_ballot_w64(any) != 0ull) {   // wave-uniform
                    IBGS_LANES_ADD(1, 1);
                    // v[]: 0 Sx, 1 Sy, 2 Ax, 3 Ay, 4 Sxx, 5 Sxy, 6 Syy, 7 S0, 8-10 rgb (= grad_acc columns)
                    float v[12];
                    if (!RISK && ((riskm >> j) & 1ull)) l_moments<PPL>(v, Q, dx0, dy0, ca, cb, cc);          // wave-uniform: a near-singular conic (see above)
                    else
                    if constexpr (PPL == 4) {
                        const float A = Q[1] + Q[3], B = Q[2] + Q[3], S0 = (Q[0] + Q[2]) + A;
                        const float t = dx0 * S0, u = dy0 * S0;
                        v[0] = fmaf(-8.0f, A, t); v[1] = fmaf(-8.0f, B, u);
                        v[4] = fmaf(dx0, fmaf(-16.0f, A, t), 64.0f * A);
                        v[6] = fmaf(dy0, fmaf(-16.0f, B, u), 64.0f * B);
                        v[5] = fmaf(-8.0f, fmaf(-8.0f, Q[3], dx0 * B), dy0 * v[0]);
                        v[7] = S0;
                    } else {
                        const float qdx = Q[0] * dx0, qdy = Q[0] * dy0;
                        v[0] = qdx; v[1] = qdy; v[4] = qdx * dx0; v[5] = qdx * dy0; v[6] = qdy * dy0; v[7] = Q[0];
                    }
                    v[2] = aX; v[3] = aY;          // (conic * d was formed with the scaled conic: preprocess_bwd multiplies these two sums by EXP2_UNSCALE)
                    v[8] = vR; v[9] = vG; v[10] = vB; v[11] = 0.f;
                    const float tot = wave_transpose_reduce12(v, lane);
                    const uint32_t id = __float_as_uint(q0.w);
                    if (col >= 0) {
                        if (p.slab) p.slab[((size_t)(r0 + k) * (p.slab_ipt ? p.slab_ipt : IPT) + (size_t)sub) * GACC_FLOATS + col] = tot;      // wave-uniform choice
                        else atomicAdd(p.gacc + (size_t)id * GACC_FLOATS + col, tot);
                    }
                }
            }
        };
        if (bg0) { if (stable) chunk(std::true_type{}, std::true_type{}); else chunk(std::false_type{}, std::true_type{}); }
        else { if (stable) chunk(std::true_type{}, std::false_type{}); else chunk(std::false_type{}, std::false_type{}); }
        __syncthreads();
        top -= count;
    }
    IBGS_LANES_FLUSH(g_lanes_bwd);
}

// ---- geo variant, pass 1: the median / warp terms of every buffered contributor, per pixel ---------------------------------
// The reference evaluates the block of backward.cu:692-771 inside the blend loop for every accepted pair whose list position lies
// in the pixel's window [min - 1, max - 1] with a positive ray/plane depth -- exactly the contributors the forward buffered (the
// last ceil(L/2) before T crosses 0.5, the first floor(L/2) after it).  Everything in that block except the blend
// weight w = alpha T of the pair is a function of (pixel, Gaussian) alone, and w enters linearly.  So one thread per PIXEL
// computes, for each of its <= L buffered contributors (numbers saved by the forward, img arena `slot_c`):
//     E   what the block adds to dL/dalpha:  dL/dmedian (d - median) / sum_w + sum_m dL/dwarped[m] . (c_m - warped[m]) / sw_m
//     K   what it adds to dL/d(plane parameters) PER UNIT of w (with the cumulative depth gradient of quirk Q2 already summed
//         over the in-bounds sources):  Kx, Ky, Kz for the normal, Kd for the distance
// and writes them sorted by contributor number, descending = the order the back-to-front traversal meets them, closed by a zero.
// The blend loop (pass 2) then handles a window pair with one compare, five loads and five fma instead of ~300 instructions and
// 25 texel gathers under a one-lane exec mask, and no longer carries the warp code's registers.
// Table layout: tab[(slot * 6 + field) * HW + pixel], fields = {contributor number (uint bits), E, Kx, Ky, Kz, Kd}.
// (six waves per SIMD: the compiler settles at 101 VGPRs = 4 waves when left alone; at 80 -- no spills -- the pass takes 0.205 instead of 0.219 ms on the trained scene,
// 0.226 instead of 0.239 at C3; 72 VGPRs / 7 waves spill 36 bytes per lane: 0.239; 64 / 8: 0.434.  Fewer instructions at the price of registers -- 9 gathers instead of 13,
// reciprocals instead of IEEE divisions -- lost every time: the pass lives on the waves it can keep in flight)
__global__ void __launch_bounds__(256, 6) geo_window_kernel(BwdParams p)
{
    // no fused multiply-adds: the in-bounds test of backward.cu:722 is a decision on a projected coordinate, which is then the oracle's
    // bit for bit (the depth it starts from is evaluated in double by both); the pass is bound by its gathers
#pragma clang fp contract(off)
    const int W = p.cam.W, H = p.cam.H;
    const size_t HW = (size_t)W * H;
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int px = (int)(pix % W), py = (int)(pix / W);
    const float pxf = (float)px, pyf = (float)py;
    const float fx = p.cam.fx, fy = p.cam.fy;
    const float cx = (float)(W * 0.5f), cy = (float)(H * 0.5f);
    // the forward's buffer_length, but never more entries than the caller's table holds (a caller that states a smaller buffer_length than its
    // forward used gets truncated windows, not a write past its scratch: include/ibgs_rast.h)
    const int L = min((int)p.meta[0], p.tab_slots == IBGS_MAX_BUFFER_LENGTH ? IBGS_MAX_BUFFER_LENGTH : p.tab_slots - 1);          // (a full table of 8 slots holds 8 entries and needs no terminator)
    float* tab = p.tab + pix;
    auto put = [&](int slot, int field, float v) { tab[(size_t)(slot * GEO_TAB_FIELDS + field) * HW] = v; };

    // buffered contributors, sorted by number, descending (L <= 8: insertion sort in registers)
    uint32_t cs[IBGS_MAX_BUFFER_LENGTH];
    int ns = 0;
    const uint32_t lo = p.low_high[2 * pix];
    if (lo != 0u) {           // Q4: slot 0 empty -> min contributor 0 -> the unsigned window test never passes: no entry for this pixel
        for (int s = 0; s < L; s++) {
            const uint32_t c = p.slot_c[(size_t)s * HW + pix];
            if (c == 0u) continue;
            int at = ns++;
#pragma unroll
            for (int k = IBGS_MAX_BUFFER_LENGTH - 1; k > 0; k--) if (k <= at && cs[k - 1] < c) { cs[k] = cs[k - 1]; at = k - 1; }
            cs[at] = c;
        }
    }
    if (ns == 0) { put(0, 0, __uint_as_float(0u)); return; }

    const int tile = (py / TILE) * p.cam.gx + px / TILE;
    const uint32_t r0 = p.ranges[2 * tile];
    // dbl: backward.cu:545 evaluates (pix - W*0.5)/fx in double
    const float rayx = (float)(((double)pxf - W * 0.5) / (double)fx), rayy = (float)(((double)pyf - H * 0.5) / (double)fy);
    const float sumw = p.sum_w[pix];
    const float g_d = p.dL_ddepth ? p.dL_ddepth[pix] : 0.f;
    const float med = p.depth_pixels[pix];
    int out = 0;
    for (int i = 0; i < ns; i++) {
        const uint32_t c = cs[i];
        const uint32_t gid = p.point_list[r0 + c - 1u];
        const float4 q1 = p.rec[(size_t)gid * 4 + 1], q3 = p.rec[(size_t)gid * 4 + 3];
        const float dist = q1.w;
        const float dotn = q3.x * rayx + q3.y * rayy + q3.z;
        const float tmp = (float)((double)dotn + 1.0e-8);                 // dbl, backward.cu:697
        const float tmp2 = dist / (tmp * tmp);
        const float dep = (float)(-(double)dist / ((double)dotn + 1.0e-8));  // dbl, backward.cu:699
        if (!(dep > 0.0f)) continue;            // (cannot happen for a buffered contributor; kept as the reference's gate)
        const float X = (pxf - cx) * dep / fx, Y = (pyf - cy) * dep / fy, Z = dep;
        float kd = g_d / sumw;                  // depth gradient per unit of w
        float E = g_d * (dep - med) / sumw;
        float Kd = 0.f, Kx = 0.f, Ky = 0.f, Kz = 0.f;
        for (int m = 0; m < IBGS_MAX_SRC; m++) {
            const int si = p.valid_idx[(size_t)m * HW + pix];
            if (si == -1) break;
            const float* r = p.ref_to_src + 16 * si;
            const float tx = r[0] * X + r[1] * Y + r[2] * Z + r[3];
            const float ty = r[4] * X + r[5] * Y + r[6] * Z + r[7];
            const float tz = r[8] * X + r[9] * Y + r[10] * Z + r[11];
            const float u = (tx * fx / tz) + cx, vv_ = (ty * fy / tz) + cy;
            if (u >= 0 && u <= W - 1 && vv_ >= 0 && vv_ <= H - 1) {
                // past the decision: values only (gradient terms, compared with the oracle to a tolerance) -- multiply-adds may fuse here, which takes a
                // third of the block's ~140 instructions away (tex_rgba_b keeps its own setting: the forward's fetch, bit for bit)
#pragma clang fp contract(fast)
                const float4* img = p.src_rgba + (size_t)si * HW;
                const float4 c4 = tex_rgba_b(img, W, H, u + 0.5f, vv_ + 0.5f, p.tex_quant);
                const float cc3[3] = {c4.x, c4.y, c4.z};
                const float sw = p.valid_w[(size_t)m * HW + pix];
                float gc[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const float gw = p.dL_dwarped ? p.dL_dwarped[((size_t)m * 3 + ch) * HW + pix] : 0.f;
                    gc[ch] = gw / sw;
                    E += gw * (cc3[ch] - p.warped_pixels[((size_t)m * 3 + ch) * HW + pix]) / sw;
                }
                const float Av = (pxf - cx) / fx, Bv = (pyf - cy) / fy;
                const float U = r[0] * Av + r[1] * Bv + r[2];
                const float V = r[4] * Av + r[5] * Bv + r[6];
                const float Wc = r[8] * Av + r[9] * Bv + r[10];
                const float den = (Wc * dep + r[11]);
                const float dpx = fx * (U * r[11] - Wc * r[3]) / (den * den);
                const float dpy = fy * (V * r[11] - Wc * r[7]) / (den * den);
                // four linear-filtered fetches at integer coordinates (u0, v0), (u0 + 1, v0), (u0, v0 + 1), (u0 + 1, v0 + 1).  With
                // texel centres at i + 0.5 each of them averages the 2 x 2 texels around its corner (weights 1/4 each), and the four
                // corners share the 3 x 3 block of texels u0 - 1 .. u0 + 1, v0 - 1 .. v0 + 1 (clamped): NINE loads instead of sixteen, the
                // same operations in the same order per fetch (no fused multiply-adds, like tex_rgba_b and the oracle's tex_rgb).  This pass
                // runs at the L1 bandwidth of the chip (16-byte gathers: 20 per (buffered contributor, source) before, 13 now).
                const float uu = u + 0.5f, vv2 = vv_ + 0.5f;
                const int u0 = (int)floorf(uu), v0 = (int)floorf(vv2);
                const float fu = uu - (float)u0, fv = vv2 - (float)v0, fu1 = 1.0f - fu, fv1 = 1.0f - fv;
                float4 I00, I01, I10, I11;
                {
                    const int xi0 = min(W - 1, max(0, u0 - 1)), xi1 = min(W - 1, max(0, u0)), xi2 = min(W - 1, max(0, u0 + 1));
                    const int yj0 = min(H - 1, max(0, v0 - 1)), yj1 = min(H - 1, max(0, v0)), yj2 = min(H - 1, max(0, v0 + 1));
                    const float4 t00 = img[(size_t)yj0 * W + xi0], t10 = img[(size_t)yj0 * W + xi1], t20 = img[(size_t)yj0 * W + xi2];
                    const float4 t01 = img[(size_t)yj1 * W + xi0], t11 = img[(size_t)yj1 * W + xi1], t21 = img[(size_t)yj1 * W + xi2];
                    const float4 t02 = img[(size_t)yj2 * W + xi0], t12 = img[(size_t)yj2 * W + xi1], t22 = img[(size_t)yj2 * W + xi2];
                    // tex_rgba_b at an integer coordinate: a = b = 0.5 exactly (also after the 8-bit weight rounding), all four weights 0.25
                    const float a = quant8b(0.5f, p.tex_quant), b = quant8b(0.5f, p.tex_quant);
                    const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
                    auto quad = [&](const float4& q00, const float4& q10, const float4& q01, const float4& q11) {
                        float4 r;
                        r.x = w00 * q00.x + w10 * q10.x + w01 * q01.x + w11 * q11.x;
                        r.y = w00 * q00.y + w10 * q10.y + w01 * q01.y + w11 * q11.y;
                        r.z = w00 * q00.z + w10 * q10.z + w01 * q01.z + w11 * q11.z;
                        r.w = 1.0f;
                        return r;
                    };
                    I00 = quad(t00, t10, t01, t11);          // corner (u0, v0): texels u0 - 1 .. u0, v0 - 1 .. v0
                    I01 = quad(t10, t20, t11, t21);          // corner (u0 + 1, v0)
                    I10 = quad(t01, t11, t02, t12);          // corner (u0, v0 + 1)
                    I11 = quad(t11, t21, t12, t22);          // corner (u0 + 1, v0 + 1)
                }
                const float dIu0 = -fv1 * I00.x + fv1 * I01.x - fv * I10.x + fv * I11.x;
                const float dIu1 = -fv1 * I00.y + fv1 * I01.y - fv * I10.y + fv * I11.y;
                const float dIu2 = -fv1 * I00.z + fv1 * I01.z - fv * I10.z + fv * I11.z;
                const float dIv0 = -fu1 * I00.x - fu * I01.x + fu1 * I10.x + fu * I11.x;
                const float dIv1 = -fu1 * I00.y - fu * I01.y + fu1 * I10.y + fu * I11.y;
                const float dIv2 = -fu1 * I00.z - fu * I01.z + fu1 * I10.z + fu * I11.z;
                const float du = gc[0] * dIu0 + gc[1] * dIu1 + gc[2] * dIu2;
                const float dv = gc[0] * dIv0 + gc[1] * dIv1 + gc[2] * dIv2;
                kd += du * dpx + dv * dpy;
                // the plane parameters receive the depth gradient inside the per-source in-bounds branch, cumulatively
                Kd += (-kd / tmp);
                Kx += kd * tmp2 * rayx;
                Ky += kd * tmp2 * rayy;
                Kz += kd * tmp2;
            }
        }
        put(out, 0, __uint_as_float(c)); put(out, 1, E); put(out, 2, Kx); put(out, 3, Ky); put(out, 4, Kz); put(out, 5, Kd);
        out++;
    }
    if (out < p.tab_slots) put(out, 0, __uint_as_float(0u));
}

// ---- geo variant, pass 2: the blend loop -----------------------------------------------------------------------------------
// render_bwd_color_body plus (i) the three normal channels, blended like colour channels (they share S), (ii) the window pairs:
// a pixel's next buffered contributor number sits in a register; when the traversal reaches it (one integer compare per quadrant
// and Gaussian) the lane loads that entry of the window table, adds E to dL/dalpha and w K to the plane sums, and moves on.
template <int PPL, bool ABS = true, bool RISK = false>          // ABS = false: IBGS_FLAG_NO_ABS_GRAD; RISK: a flagged tile -- both as in the colour body
__device__ __forceinline__ void render_bwd_geo_body(const BwdParams& p, const int tile, const int sub, float4 (&s_rec)[4][BWD_CHUNK],
                                                    float4 (*s_gpix)[WAVE] /* PPL rows: dL/dC (rgb), -T_final * (bg . dL/dC) */, float4 (*s_gnrm)[WAVE] /* PPL rows: dL/dN (xyz) */)
{
    constexpr int CHUNK = BWD_CHUNK;          // (the LDS is the caller's, as for the colour body)
    const int lane = threadIdx.x;
    int col = reduce16_column(lane);
    if (col >= 15) col = -1;
    constexpr int IPT = 4 / PPL;
    const int quad0 = sub * PPL;
    const int W = p.cam.W, H = p.cam.H;
    const int tx0 = (tile % p.cam.gx) * TILE, ty0 = (tile / p.cam.gx) * TILE;
    const size_t HW = (size_t)W * H;

    float T[PPL], S[PPL];
    const bool bg0 = p.cam.bg[0] == 0.f && p.cam.bg[1] == 0.f && p.cam.bg[2] == 0.f;      // wave-uniform (scalar loads), as in the colour body
    uint32_t ncontrib[PPL], next_c[PPL], slot[PPL], pixo[PPL];
    uint32_t nmax = 0;
    const float pxf0 = (float)(tx0 + (quad0 & 1) * 8 + (lane & 7)), pyf0 = (float)(ty0 + (quad0 >> 1) * 8 + (lane >> 3));
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        const int qq = quad0 + q;
        const int px = tx0 + (qq & 1) * 8 + (lane & 7), py = ty0 + (qq >> 1) * 8 + (lane >> 3);
        const bool inside = px < W && py < H;
        const size_t pix = (size_t)py * W + px;
        pixo[q] = inside ? (uint32_t)pix : 0u;
        const float T_final = inside ? p.final_T[pix] : 0.f;
        T[q] = T_final;
        ncontrib[q] = inside ? p.n_contrib[pix] : 0u;
        nmax = max(nmax, ncontrib[q]);
        S[q] = 0.f;
        float4 g, gn;
        g.x = (inside && p.dL_dcolor) ? p.dL_dcolor[pix] : 0.f;
        g.y = (inside && p.dL_dcolor) ? p.dL_dcolor[HW + pix] : 0.f;
        g.z = (inside && p.dL_dcolor) ? p.dL_dcolor[2 * HW + pix] : 0.f;
        g.w = -T_final * (p.cam.bg[0] * g.x + p.cam.bg[1] * g.y + p.cam.bg[2] * g.z);
        gn.x = (inside && p.dL_dnormal) ? p.dL_dnormal[pix] : 0.f;
        gn.y = (inside && p.dL_dnormal) ? p.dL_dnormal[HW + pix] : 0.f;
        gn.z = (inside && p.dL_dnormal) ? p.dL_dnormal[2 * HW + pix] : 0.f;
        gn.w = 0.f;
        s_gpix[q][lane] = g; s_gnrm[q][lane] = gn;
        slot[q] = 0;
        next_c[q] = (inside && p.tab) ? __float_as_uint(p.tab[pix]) : 0u;          // table entry 0, field 0 (0 = no window pair at all; no table: no depth / warp gradient came in)
    }
    nmax = wave_max_u32(nmax);
    const uint32_t r0 = p.ranges[2 * tile], r1 = p.ranges[2 * tile + 1];
    const int n = (int)(r1 - r0);
    int top = __builtin_amdgcn_readfirstlane(min((int)nmax, n));

    while (top > 0) {
        const int count = min(CHUNK, top);
        bool risky = false;
        if (lane < count) {
            const uint32_t id = p.point_list[r0 + (uint32_t)(top - 1 - lane)];
            const float4* r = p.rec + (size_t)id * 4;
            float4 ra = r[0];
            ra.w = __uint_as_float(id);
            float4 c1 = r[1];
            risky = conic_takes_ref_power(c1.x, c1.y, c1.z);
            stage_for_exp2(ra, c1);
            s_rec[0][lane] = ra; s_rec[1][lane] = c1; s_rec[2][lane] = r[2]; s_rec[3][lane] = r[3];
        }
        const uint64_t riskm = p.power_skip ? __builtin_amdgcn_ballot_w64(risky) : 0ull;
        uint64_t ncm[PPL];
        bool stable = true;
#pragma unroll
        for (int q = 0; q < PPL; q++) {
            ncm[q] = __builtin_amdgcn_ballot_w64((uint32_t)(top - count) < ncontrib[q]);
            stable = stable && (ncm[q] == __builtin_amdgcn_ballot_w64((uint32_t)(top - 1) < ncontrib[q]));
        }
        __syncthreads();
        auto chunk = [&](auto stable_tag, auto bg0_tag) {
            constexpr bool STABLE = decltype(stable_tag)::value;
            constexpr bool BG0 = decltype(bg0_tag)::value;          // black background: the -T_final (bg . g) / (1 - alpha) term is zero
            for (int j = 0; j < count; j++) {
                const uint32_t k = (uint32_t)(top - 1 - j);
                const float4 q0 = s_rec[0][j], q1 = s_rec[1][j], q2 = s_rec[2][j], q3 = s_rec[3][j];
                const float ca = q1.x, cb = q1.y, cc = q1.z, nlo = q0.z;          // scaled conic, -log2(opacity)
                const float dx0 = q0.x - pxf0, dy0 = q0.y - pyf0;
                const float lx0 = ca * dx0 + cb * dy0, ly0 = cb * dx0 + cc * dy0;
                const float P0 = fmaf(dx0, lx0, fmaf(dy0, ly0, nlo));          // the forward's E (render_fwd.hip)
                float p2q[PPL], lxq[PPL], lyq[PPL];
                p2q[0] = P0; lxq[0] = lx0; lyq[0] = ly0;
                if constexpr (PPL >= 2) {
                    p2q[1] = fmaf(-16.0f, lx0, P0 + 64.0f * ca);
                    if constexpr (ABS) { lxq[1] = fmaf(-8.0f, ca, lx0); lyq[1] = fmaf(-8.0f, cb, ly0); }
                }
                if constexpr (PPL == 4) {
                    p2q[2] = fmaf(-16.0f, ly0, P0 + 64.0f * cc);
                    p2q[3] = fmaf(128.0f, cb, p2q[2] + (p2q[1] - P0));          // E3 = E2 + (E1 - E0) + 128 b: three instructions, as the forward
                    if constexpr (ABS) {
                        lxq[2] = fmaf(-8.0f, cb, lx0); lyq[2] = fmaf(-8.0f, cc, ly0);
                        lxq[3] = fmaf(-8.0f, cb, lxq[1]); lyq[3] = fmaf(-8.0f, cc, lyq[1]);
                    }
                }
                if constexpr (!RISK) {
                    if ((riskm >> j) & 1ull) {          // wave-uniform and rare: E from the reference's expression, `power > 0` pairs dropped (common.h); the sums: l_moments below
                        const uint32_t gid = __builtin_amdgcn_readfirstlane(__float_as_uint(q0.w));
                        const float4 g0 = p.rec[(size_t)gid * 4], g1 = p.rec[(size_t)gid * 4 + 1];
#pragma unroll
                        for (int q = 0; q < PPL; q++) {
                            const float ox = (PPL == 4) ? (float)((q & 1) * 8) : (float)(q * 8), oy = (PPL == 4) ? (float)((q >> 1) * 8) : 0.f;
                            p2q[q] = ref_power_E(g0.x - (pxf0 + ox), g0.y - (pyf0 + oy), g1.x, g1.y, g1.z, nlo);
                        }
                    }
                }
                if constexpr (RISK) {
                    if ((riskm >> j) & 1ull) {          // wave-uniform: a near-singular conic in the reference's arithmetic, as in the colour body (+ the normal channels and the window pairs)
                        const uint32_t gid = __builtin_amdgcn_readfirstlane(__float_as_uint(q0.w));
                        const float4 g0 = p.rec[(size_t)gid * 4], g1 = p.rec[(size_t)gid * 4 + 1];
                        float rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rR = 0.f, rG = 0.f, rB = 0.f, rNx = 0.f, rNy = 0.f, rNz = 0.f, rD = 0.f;
                        bool rany = false;
#pragma unroll
                        for (int q = 0; q < PPL; q++) {
                            const float ox = (PPL == 4) ? (float)((q & 1) * 8) : (float)(q * 8), oy = (PPL == 4) ? (float)((q >> 1) * 8) : 0.f;
                            const float dx = g0.x - (pxf0 + ox), dy = g0.y - (pyf0 + oy);
                            const float power = ref_power(dx, dy, g1.x, g1.y, g1.z);
                            const float E = ref_power_E(dx, dy, g1.x, g1.y, g1.z, nlo);
                            const uint64_t okm = __builtin_amdgcn_ballot_w64(k < ncontrib[q]) & __builtin_amdgcn_ballot_w64(E <= ALPHA_SKIP_E);
                            if (okm == 0ull) continue;
                            rany = true;
                            const float Gr = __builtin_amdgcn_inverse_ballot_w64(okm) ? expf(power) : 0.f;
                            const float oG = g0.z * Gr;
                            const float alpha = fminf(0.99f, oG);
                            const float om = 1.f - alpha;
                            T[q] = T[q] / om;
                            const float w = alpha * T[q];
                            const float4 gp = s_gpix[q][lane], gn = s_gnrm[q][lane];
                            const float cg = q2.x * gp.x + q2.y * gp.y + q2.z * gp.z + q3.x * gn.x + q3.y * gn.y + q3.z * gn.z;
                            float dL_dalpha = cg - S[q];
                            S[q] = fmaf(alpha, dL_dalpha, S[q]);
                            rR = fmaf(w, gp.x, rR); rG = fmaf(w, gp.y, rG); rB = fmaf(w, gp.z, rB);
                            rNx = fmaf(w, gn.x, rNx); rNy = fmaf(w, gn.y, rNy); rNz = fmaf(w, gn.z, rNz);
                            const uint64_t hit = okm & __builtin_amdgcn_ballot_w64(k + 1u == next_c[q]);          // window pair (as on the fast path below)
                            if (hit != 0ull) {
                                if (__builtin_amdgcn_inverse_ballot_w64(hit)) {
                                    const float* e = p.tab + (size_t)(slot[q] * GEO_TAB_FIELDS) * HW + pixo[q];
                                    dL_dalpha += e[HW];
                                    rNx = fmaf(w, e[2 * HW], rNx); rNy = fmaf(w, e[3 * HW], rNy); rNz = fmaf(w, e[4 * HW], rNz);
                                    rD = fmaf(w, e[5 * HW], rD);
                                    slot[q]++;
                                    next_c[q] = (slot[q] < (uint32_t)p.tab_slots) ? __float_as_uint(e[(size_t)GEO_TAB_FIELDS * HW]) : 0u;
                                }
                            }
                            dL_dalpha = dL_dalpha * T[q];
                            if constexpr (!BG0) dL_dalpha += gp.w / om;
                            ref_pair_sums(rs, Gr, dL_dalpha, dx, dy, g0.z, g1.x, g1.y, g1.z);
                        }
                        if (__builtin_amdgcn_ballot_w64(rany) != 0ull) {
                            float v[16] = {rs[0], rs[1], rs[2], rs[3], rs[4], rs[5], rs[6], rs[7], rR, rG, rB, rNx, rNy, rNz, rD, 0.f};
                            const float tot = wave_transpose_reduce16(v, lane);
                            if (col >= 0) {
                                if (p.slab) p.slab[((size_t)(r0 + k) * (p.slab_ipt ? p.slab_ipt : IPT) + (size_t)sub) * GACC_FLOATS + col] = tot;
                                else atomicAdd(p.gacc + (size_t)gid * GACC_FLOATS + col, tot);
                            }
                        }
                        continue;
                    }
                }
                float Q[PPL], aX = 0.f, aY = 0.f, vR = 0.f, vG = 0.f, vB = 0.f, vNx = 0.f, vNy = 0.f, vNz = 0.f, vD = 0.f;
                bool any = false;
#pragma unroll
                for (int q = 0; q < PPL; q++) {
                    const uint64_t live = STABLE ? ncm[q] : __builtin_amdgcn_ballot_w64(k < ncontrib[q]);
                    const uint64_t okm = live & __builtin_amdgcn_ballot_w64(p2q[q] <= ALPHA_SKIP_E);
                    Q[q] = 0.f;
                    if (okm != 0ull) {
                        any = true;
                        const float oG = select_or_zero_after_trans(okm, __builtin_amdgcn_exp2f(-p2q[q]));
                        const float alpha = min_099(oG);
                        const float rinv = __builtin_amdgcn_rcpf(1.f - alpha);
                        T[q] = T[q] * rinv;
                        const float w = alpha * T[q];
                        const float4 gp = s_gpix[q][lane], gn = s_gnrm[q][lane];
                        // the normal channels are blended like three more colour channels: they share S
                        const float cg = q2.x * gp.x + q2.y * gp.y + q2.z * gp.z + q3.x * gn.x + q3.y * gn.y + q3.z * gn.z;
                        float dL_dalpha = cg - S[q];
                        S[q] = fmaf(alpha, dL_dalpha, S[q]);
                        vR = fmaf(w, gp.x, vR); vG = fmaf(w, gp.y, vG); vB = fmaf(w, gp.z, vB);
                        vNx = fmaf(w, gn.x, vNx); vNy = fmaf(w, gn.y, vNy); vNz = fmaf(w, gn.z, vNz);
                        // window pair?  (a buffered contributor is an accepted pair by construction; the test is taken under okm anyway)
                        const uint64_t hit = okm & __builtin_amdgcn_ballot_w64(k + 1u == next_c[q]);
                        if (hit != 0ull) {                                           // wave-uniform
                            if (__builtin_amdgcn_inverse_ballot_w64(hit)) {
                                const float* e = p.tab + (size_t)(slot[q] * GEO_TAB_FIELDS) * HW + pixo[q];
                                dL_dalpha += e[HW];
                                vNx = fmaf(w, e[2 * HW], vNx); vNy = fmaf(w, e[3 * HW], vNy); vNz = fmaf(w, e[4 * HW], vNz);
                                vD = fmaf(w, e[5 * HW], vD);
                                slot[q]++;
                                next_c[q] = (slot[q] < (uint32_t)p.tab_slots) ? __float_as_uint(e[(size_t)GEO_TAB_FIELDS * HW]) : 0u;
                            }
                        }
                        if constexpr (BG0) dL_dalpha = dL_dalpha * T[q];
                        else dL_dalpha = fmaf(dL_dalpha, T[q], gp.w * rinv);
                        const float qv = oG * dL_dalpha;
                        Q[q] = qv;
                        if constexpr (ABS) { aX = fmaf(fabsf(qv), fabsf(lxq[q]), aX); aY = fmaf(fabsf(qv), fabsf(lyq[q]), aY); }      // |a b| = |a| |b|: one v_fma with source modifiers
                    }
                }
                if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
                    // v[]: 0 Sx, 1 Sy, 2 Ax, 3 Ay, 4 Sxx, 5 Sxy, 6 Syy, 7 S0, 8-10 rgb, 11-13 normal, 14 dist (= grad_acc columns)
                    float v[16];
                    if (!RISK && ((riskm >> j) & 1ull)) l_moments<PPL>(v, Q, dx0, dy0, ca, cb, cc);          // wave-uniform: a near-singular conic (see above)
                    else
                    if constexpr (PPL == 4) {
                        const float A = Q[1] + Q[3], B = Q[2] + Q[3], S0 = (Q[0] + Q[2]) + A;
                        const float t = dx0 * S0, u = dy0 * S0;
                        v[0] = fmaf(-8.0f, A, t); v[1] = fmaf(-8.0f, B, u);
                        v[4] = fmaf(dx0, fmaf(-16.0f, A, t), 64.0f * A);
                        v[6] = fmaf(dy0, fmaf(-16.0f, B, u), 64.0f * B);
                        v[5] = fmaf(-8.0f, fmaf(-8.0f, Q[3], dx0 * B), dy0 * v[0]);
                        v[7] = S0;
                    } else if constexpr (PPL == 2) {       // two quadrants side by side: d1 = d0 - (8, 0)
                        const float S0 = Q[0] + Q[1];
                        const float t = dx0 * S0;
                        v[0] = fmaf(-8.0f, Q[1], t); v[1] = dy0 * S0;
                        v[4] = fmaf(dx0, fmaf(-16.0f, Q[1], t), 64.0f * Q[1]);
                        v[5] = dy0 * v[0]; v[6] = dy0 * v[1];
                        v[7] = S0;
                    } else {
                        const float qdx = Q[0] * dx0, qdy = Q[0] * dy0;
                        v[0] = qdx; v[1] = qdy; v[4] = qdx * dx0; v[5] = qdx * dy0; v[6] = qdy * dy0; v[7] = Q[0];
                    }
                    v[2] = aX; v[3] = aY;          // (conic * d was formed with the scaled conic: preprocess_bwd multiplies these two sums by EXP2_UNSCALE)
                    v[8] = vR; v[9] = vG; v[10] = vB; v[11] = vNx; v[12] = vNy; v[13] = vNz; v[14] = vD; v[15] = 0.f;
                    const float tot = wave_transpose_reduce16(v, lane);
                    const uint32_t id = __float_as_uint(q0.w);
                    if (col >= 0) {
                        if (p.slab) p.slab[((size_t)(r0 + k) * (p.slab_ipt ? p.slab_ipt : IPT) + (size_t)sub) * GACC_FLOATS + col] = tot;
                        else atomicAdd(p.gacc + (size_t)id * GACC_FLOATS + col, tot);
                    }
                }
            }
        };
        if (bg0) { if (stable) chunk(std::true_type{}, std::true_type{}); else chunk(std::false_type{}, std::true_type{}); }
        else { if (stable) chunk(std::true_type{}, std::false_type{}); else chunk(std::false_type{}, std::false_type{}); }
        __syncthreads();
        top -= count;
    }
}

// One entry point per variant so that each gets its own register budget.  Large frames: one wave per tile; small frames (fewer
// tiles than wave slots): one wave per 8x8 quadrant so that the chip fills up.
#ifdef IBGS_TRACE_WAVES
__device__ uint4 g_trace_bwd[IBGS_TRACE_MAX];
#endif
__global__ void __launch_bounds__(64, 8) render_bwd_color_kernel(BwdParams p)
{
    IBGS_TRACE_BEGIN();
    int tile, sub = 0;
    bool have;
    if (p.order) { const uint32_t t = p.order[blockIdx.x]; tile = (int)(t & ~ORDER_SPLIT_BIT); have = t != 0xFFFFFFFFu; }
    else have = tile_map_item(p.tmap, blockIdx.x, p.cam.gx, p.cam.gy, 1, tile, sub);
    __shared__ float4 s_rec[3][BWD_CHUNK];
    __shared__ float4 s_gpix[4][WAVE];
    if (have && !tile_is_risky(p.tile_risky, tile)) render_bwd_color_body<4>(p, tile, sub, s_rec, s_gpix);
    IBGS_TRACE_END(g_trace_bwd);
}
__global__ void __launch_bounds__(64, 8) render_bwd_color_noabs_kernel(BwdParams p)          // IBGS_FLAG_NO_ABS_GRAD
{
    int tile, sub = 0;
    bool have;
    if (p.order) { const uint32_t t = p.order[blockIdx.x]; tile = (int)(t & ~ORDER_SPLIT_BIT); have = t != 0xFFFFFFFFu; }
    else have = tile_map_item(p.tmap, blockIdx.x, p.cam.gx, p.cam.gy, 1, tile, sub);
    __shared__ float4 s_rec[3][BWD_CHUNK];
    __shared__ float4 s_gpix[4][WAVE];
    if (have && !tile_is_risky(p.tile_risky, tile)) render_bwd_color_body<4, false>(p, tile, sub, s_rec, s_gpix);
}
// The flagged tiles (their list holds a near-singular conic), one wave per tile whatever the frame's size, in the order of the kernel it is launched beside
// (p.order) or in tile order; every other workgroup leaves at once.  128 VGPRs: the reference-arithmetic block spills nothing.
template <bool ABS>
__global__ void __launch_bounds__(64, 4) render_bwd_color_risk_kernel(BwdParams p)
{
    __shared__ float4 s_rec[3][BWD_CHUNK];
    __shared__ float4 s_gpix[4][WAVE];
    int tile = (int)blockIdx.x;
    if (p.order) { const uint32_t t = p.order[blockIdx.x]; if (t == 0xFFFFFFFFu) return; tile = (int)(t & ~ORDER_SPLIT_BIT); }
    if (tile >= p.ntiles || !tile_is_risky(p.tile_risky, tile)) return;
    render_bwd_color_body<4, ABS, true>(p, tile, 0, s_rec, s_gpix);
}

// ---- balanced launch order for the colour kernel ----------------------------------------------------------------------------------
// One wave per tile, all of them resident at once (8 160 tiles on 8 192 slots at 1080p): the dispatcher puts workgroups i and i + 1024 on the
// same SIMD (measured: tools/wave_trace.py --placement; XCD = i % 8), so a SIMD's load is the sum of "its" eight tiles and the kernel ends
// with the heaviest SIMD.  In tile order that sum varies like eight random tiles: the SIMDs end at 94 % of the span on average with C3's
// uniform lists, at 83 % with trained-like opacities (cv of the work per tile 0.06 / 0.21) -- tools/balance_stats.py models it, the wave
// stamps confirm it (92 % / 82 %).  The work of a tile is known: the forward wrote how far it walked every list (ImgState::tile_walked).
// This kernel sorts the tiles by it, descending (counting sort on the top 10 bits, one workgroup), and deals the ranks out in SNAKE order
// over the 1 024 SIMD classes: stratum r / 1024 of rank r goes to one round, class r % 1024 forwards for even strata and backwards for odd ones -- every
// class gets one tile of each octile, heavy ones paired with light ones: heaviest SIMD / mean 1.06 -> 1.01 (uniform), 1.20 -> 1.02 (trained).
// The tile map plays no role for this kernel (profiles/r03_tile_map.txt: every layout within 1 %).  Frames with more tiles than slots are launched
// in plain descending order (later workgroups start as slots free up: longest first is what a queue wants).
// A pure performance heuristic: any order gives the same gradients (the deterministic mode's slab is indexed by list position).
constexpr int ORDER_SNAKE_ROUNDS = 8;
__global__ void __launch_bounds__(1024) tile_order_kernel(int ntiles, int nslots /* ntiles rounded up to ORDER_CLASSES */, int slot_rounds /* waves per SIMD of the kernel that follows */,
                                                          const uint32_t* __restrict__ walked_waves, const uint32_t* __restrict__ meta, uint32_t* __restrict__ order,
                                                          uint32_t* __restrict__ order_out /* the caller's copy (ibgs_backward_args::tile_order_out) or nullptr */,
                                                          int theta_pct = 0 /* > 0 (hybrid kernels): bit 31 of a tile's word says that four quadrant waves should walk it (hybrid_split, common.h) */)
{
    __shared__ uint32_t s_hist[1024];
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_max;
    __shared__ unsigned long long s_total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ipt = meta[10];          // waves per tile of the forward variant that ran (1, 2 or 4): a tile was walked as far as its farthest wave
    if (ipt != 1u && ipt != 2u && ipt != 4u) {          // (an arena no forward of this library wrote)
        // (hybrid: every tile split -- four waves each are never wrong, one wave for a tile nobody measured may be)
        for (int i = tid; i < nslots; i += 1024) { const uint32_t v = i < ntiles ? ((uint32_t)i | (theta_pct > 0 ? ORDER_SPLIT_BIT : 0u)) : 0xFFFFFFFFu; order[i] = v; if (order_out) order_out[i] = v; }
        return;
    }
    auto walked_of = [&](int t) { uint32_t v = walked_waves[(size_t)t * ipt]; for (uint32_t k = 1; k < ipt; k++) v = max(v, walked_waves[(size_t)t * ipt + k]); return v; };
    s_hist[tid] = 0u;
    if (tid == 0) { s_max = 0u; s_total = 0ull; }
    for (int i = tid; i < nslots; i += 1024) { order[i] = 0xFFFFFFFFu; if (order_out) order_out[i] = 0xFFFFFFFFu; }          // every slot empty first: a backwards round that is not full leaves its holes at ITS low end
    __syncthreads();
    constexpr int KEEP = 16;          // tiles per thread kept in registers (frames up to 16 K tiles; more: read again)
    uint32_t w[KEEP];
    uint32_t m = 0;
    unsigned long long sum = 0ull;
#pragma unroll
    for (int k = 0; k < KEEP; k++) { const int t = tid + k * 1024; w[k] = t < ntiles ? walked_of(t) : 0u; m = max(m, w[k]); sum += w[k]; }
    for (int t = tid + KEEP * 1024; t < ntiles; t += 1024) { const uint32_t v = walked_of(t); m = max(m, v); sum += v; }
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (lane == 0) atomicMax(&s_max, m);
    if (theta_pct > 0) {
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
        if (lane == 0) atomicAdd(&s_total, sum);
    }
    __syncthreads();
    const uint32_t mx = s_max;
    const unsigned long long total = s_total;
    const int sh = mx >= 1024u ? (32 - __builtin_clz(mx)) - 10 : 0;
    auto bucket = [&](uint32_t v) { return 1023u - min(v >> sh, 1023u); };          // descending
#pragma unroll
    for (int k = 0; k < KEEP; k++) if (tid + k * 1024 < ntiles) atomicAdd(&s_hist[bucket(w[k])], 1u);
    for (int t = tid + KEEP * 1024; t < ntiles; t += 1024) atomicAdd(&s_hist[bucket(walked_of(t))], 1u);
    __syncthreads();
    const uint32_t v = s_hist[tid];
    uint32_t inc = v;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int ww = 0; ww < wave; ww++) before += s_w[ww];
    s_hist[tid] = before + inc - v;
    __syncthreads();
    const uint32_t nrounds = (uint32_t)nslots / (uint32_t)ORDER_CLASSES;
    auto put = [&](int t, uint32_t key) {
        const uint32_t r = atomicAdd(&s_hist[bucket(key)], 1u);          // rank among the tiles, heaviest first (ties in any order)
        const uint32_t stratum = r / ORDER_CLASSES, c = r % ORDER_CLASSES;
        uint32_t slot = r;          // more tiles than slots: workgroups beyond the slots start as earlier ones end -- heaviest first, as a queue wants it
        if (nrounds <= (uint32_t)slot_rounds) {
            // the first 1 024 workgroups are placed differently from the rest (their SIMD's next workgroup is i + 768 or i + 512, not i + 1 024:
            // tools/wave_trace.py --slowest), so that round gets the LIGHTEST stratum -- where a wrong partner costs least -- and the snake runs
            // over rounds 1 .. 7
            const uint32_t round = (stratum + 1u) % nrounds;
            slot = round * ORDER_CLASSES + ((stratum & 1u) ? ORDER_CLASSES - 1u - c : c);
        }
        const uint32_t word = (uint32_t)t | ((theta_pct > 0 && hybrid_split(key, total, (uint32_t)theta_pct)) ? ORDER_SPLIT_BIT : 0u);
        order[slot] = word;
        if (order_out) order_out[slot] = word;
    };
#pragma unroll
    for (int k = 0; k < KEEP; k++) if (tid + k * 1024 < ntiles) put(tid + k * 1024, w[k]);
    for (int t = tid + KEEP * 1024; t < ntiles; t += 1024) put(t, walked_of(t));
}
__global__ void __launch_bounds__(64, 8) render_bwd_color_small_kernel(BwdParams p)
{
    int tile, sub;
    __shared__ float4 s_rec[3][BWD_CHUNK];
    __shared__ float4 s_gpix[1][WAVE];
    if (tile_map_item(p.tmap, blockIdx.x, p.cam.gx, p.cam.gy, 4, tile, sub) && !tile_is_risky(p.tile_risky, tile)) render_bwd_color_body<1>(p, tile, sub, s_rec, s_gpix);
}
// Hybrid (frames of fewer tiles than wave slots; render_fwd.hip has the forward's twin and the reasoning): four workgroups per tile; where the forward
// walked the tile's list further than `hybrid` per cent of a SIMD's fair share of all walks, each takes a quadrant, elsewhere the first takes the tile.
// The forward left four walk lengths per tile (meta[10] = 4; tile_order_kernel sums them and marks the tiles to split); after any other forward every tile is split.
template <bool ABS>
__global__ void __launch_bounds__(64, 8) render_bwd_color_hybrid_kernel(BwdParams p)
{
    __shared__ float4 s_rec[3][BWD_CHUNK];
    __shared__ float4 s_gpix[4][WAVE];
    int tile, sub; bool split;
    if (!hybrid_item(blockIdx.x, p.hybrid_grid1, p.order, tile, sub, split) || tile_is_risky(p.tile_risky, tile)) return;
    if (split) render_bwd_color_body<1, ABS>(p, tile, sub, s_rec, s_gpix);
    else if (sub == 0) render_bwd_color_body<4, ABS>(p, tile, 0, s_rec, s_gpix);
}
template <int PPL, bool ABS>
__device__ __forceinline__ void render_bwd_geo_entry(const BwdParams& p)
{
    __shared__ float4 s_rec[4][BWD_CHUNK];
    __shared__ float4 s_gpix[PPL][WAVE];
    __shared__ float4 s_gnrm[PPL][WAVE];
    int tile, sub;
    if (PPL == 4 && p.order) { const uint32_t t = p.order[blockIdx.x]; if (t == 0xFFFFFFFFu) return; tile = (int)(t & ~ORDER_SPLIT_BIT); sub = 0; }
    else if (!tile_map_item(p.tmap, blockIdx.x, p.cam.gx, p.cam.gy, 4 / PPL, tile, sub)) return;
    if (tile_is_risky(p.tile_risky, tile)) return;
    render_bwd_geo_body<PPL, ABS>(p, tile, sub, s_rec, s_gpix, s_gnrm);
}
template <bool ABS>          // the geo pass's flagged tiles: as render_bwd_color_risk_kernel
__global__ void __launch_bounds__(64, 3) render_bwd_geo_risk_kernel(BwdParams p)
{
    __shared__ float4 s_rec[4][BWD_CHUNK];
    __shared__ float4 s_gpix[4][WAVE];
    __shared__ float4 s_gnrm[4][WAVE];
    int tile = (int)blockIdx.x;
    if (p.order) { const uint32_t t = p.order[blockIdx.x]; if (t == 0xFFFFFFFFu) return; tile = (int)(t & ~ORDER_SPLIT_BIT); }
    if (tile >= p.ntiles || !tile_is_risky(p.tile_risky, tile)) return;
    render_bwd_geo_body<4, ABS, true>(p, tile, 0, s_rec, s_gpix, s_gnrm);
}
__global__ void __launch_bounds__(64, 4) render_bwd_geo4_kernel(BwdParams p) { render_bwd_geo_entry<4, true>(p); }
__global__ void __launch_bounds__(64, 4) render_bwd_geo4_noabs_kernel(BwdParams p) { render_bwd_geo_entry<4, false>(p); }          // IBGS_FLAG_NO_ABS_GRAD
__global__ void __launch_bounds__(64, 6) render_bwd_geo_kernel(BwdParams p) { render_bwd_geo_entry<1, true>(p); }
// ... and the geo pass's hybrid (frames of fewer tiles than wave slots): one wave per tile or four quadrant waves, by the flag in the tile's order word
template <bool ABS>
__global__ void __launch_bounds__(64, 4) render_bwd_geo_hybrid_kernel(BwdParams p)
{
    __shared__ float4 s_rec[4][BWD_CHUNK];
    __shared__ float4 s_gpix[4][WAVE];
    __shared__ float4 s_gnrm[4][WAVE];
    int tile, sub; bool split;
    if (!hybrid_item(blockIdx.x, p.hybrid_grid1, p.order, tile, sub, split) || tile_is_risky(p.tile_risky, tile)) return;
    if (split) render_bwd_geo_body<1, ABS>(p, tile, sub, s_rec, s_gpix, s_gnrm);
    else if (sub == 0) render_bwd_geo_body<4, ABS>(p, tile, 0, s_rec, s_gpix, s_gnrm);
}

// BwdParams::power_skip for a call's flags: 0 = no reference branch at all; else how a near-singular conic's pairs are summed (RA_LFORM: default, RA_ASSOC: the
// reference's association -- bit RA_ASSOC also tells preprocess_bwd which row format to expect of those Gaussians)
int render_backward_ref_arith(uint32_t flags)
{
    if (flags & IBGS_FLAG_NO_REF_POWER_SKIP) return 0;
    return RA_DECIDE | ((flags & IBGS_FLAG_REF_ARITH) ? RA_ASSOC : RA_LFORM);
}

// how many waves share one tile in the variant launch_render_backward picks (1, 2 or 4): rows per list entry of the deterministic slab
int render_backward_waves_per_tile(const ibgs_backward_args& a)
{
    const int nt = ((a.W + TILE - 1) / TILE) * ((a.H + TILE - 1) / TILE);
    if (a.render_geo) return ((a.flags & IBGS_FLAG_QUADRANT_WAVES) ? false : ((a.flags & IBGS_FLAG_TILE_WAVES) ? true : nt >= 4096)) ? 1 : 4;
    return ((a.flags & IBGS_FLAG_QUADRANT_WAVES) ? true : ((a.flags & IBGS_FLAG_TILE_WAVES) ? false : nt < hybrid_max_tiles())) ? 4 : 1;
}

int launch_render_backward(hipStream_t s, const ibgs_backward_args& a, const GeomState& g, const BinState& b,
                           const ImgState& im, const float4* src_rgba, float* slab, float* geo_tab)
{
    BwdParams p;
    p.ranges = im.ranges; p.point_list = b.point_list; p.rec = reinterpret_cast<const float4*>(g.rec);
    p.cam = make_cam(a.viewmatrix, a.projmatrix, a.campos, a.bg, a.tanfovx, a.tanfovy, a.W, a.H);
    p.ntiles = p.cam.gx * p.cam.gy;
    p.n_src = a.n_src; p.tex_quant = (a.flags & IBGS_FLAG_TEX_QUANT) ? 1 : 0;
    p.power_skip = render_backward_ref_arith(a.flags);
    p.tile_risky = (p.power_skip & RA_ASSOC) ? im.tile_risky : nullptr;          // IBGS_FLAG_REF_ARITH: the flagged tiles go to the *_risk_kernel, the fast kernels skip them
    p.ref_to_src = a.ref_to_src; p.src_rgba = src_rgba;
    p.final_T = im.final_T; p.n_contrib = im.n_contrib; p.sum_w = im.sum_w; p.low_high = im.low_high;
    p.valid_idx = im.valid_idx; p.valid_w = im.valid_w;
    p.depth_pixels = a.out_depth; p.warped_pixels = a.out_warped;
    p.dL_dcolor = a.dL_dcolor; p.dL_dnormal = a.dL_dnormal; p.dL_ddepth = a.dL_ddepth; p.dL_dwarped = a.dL_dwarped;
    p.gacc = a.grad_acc; p.slab = slab; p.order = nullptr; p.slab_ipt = 0; p.hybrid_grid1 = 0;
    p.slot_c = im.slot_c; p.meta = im.meta; p.tab = geo_tab;
    p.tab_slots = (a.buffer_length >= 1 && a.buffer_length < IBGS_MAX_BUFFER_LENGTH) ? a.buffer_length + 1 : IBGS_MAX_BUFFER_LENGTH;          // as ibgs_required_geo_table_for sizes it
    const int nt = p.ntiles;
    // as the forward (render_fwd.hip): blocks of tiles per XCD; geo 8 x 4 (fetch traffic 0.73 -> 0.35 GB, clustered image 1.92 -> 1.83 ms)
    p.tmap = a.render_geo ? TileMap{TMAP_BLOCK, 1, 8, 4} : TileMap{TMAP_BLOCK, 1, 8, 8};
    auto grid = [&](int ipt) { return dim3((unsigned)tile_map_grid(p.tmap, p.cam.gx, p.cam.gy, ipt)); };
    // the flagged tiles' kernel, launched behind whichever blend kernel ran: over the slots of its launch order, or over the tiles (no flag set: every wave leaves at once)
    auto launch_risk = [&](int nslots_or_0) {
        if (!p.tile_risky) return;          // (IBGS_FLAG_REF_ARITH only)
        BwdParams r = p;
        r.slab_ipt = render_backward_waves_per_tile(a);          // rows per list entry of the deterministic slab = the waves per tile of the kernel beside it; this one is wave 0
        r.hybrid_grid1 = 0;
        if (nslots_or_0 == 0) r.order = nullptr;
        const dim3 g((unsigned)(nslots_or_0 ? nslots_or_0 : nt));
        StageTimer t(s, IBGS_STAGE_RENDER_BWD);
        const bool noabs = (a.flags & IBGS_FLAG_NO_ABS_GRAD) != 0;
        if (a.render_geo) { if (noabs) hipLaunchKernelGGL(render_bwd_geo_risk_kernel<false>, g, dim3(64), 0, s, r); else hipLaunchKernelGGL(render_bwd_geo_risk_kernel<true>, g, dim3(64), 0, s, r); }
        else { if (noabs) hipLaunchKernelGGL(render_bwd_color_risk_kernel<false>, g, dim3(64), 0, s, r); else hipLaunchKernelGGL(render_bwd_color_risk_kernel<true>, g, dim3(64), 0, s, r); }
    };
    const bool noabs = (a.flags & IBGS_FLAG_NO_ABS_GRAD) != 0;
    const int nslots = (nt + ORDER_CLASSES - 1) / ORDER_CLASSES * ORDER_CLASSES;
    // launch order of the kernel that follows (tile_order_kernel): slot_rounds = the waves per SIMD that kernel holds; theta > 0 marks the tiles four quadrant waves should walk
    auto launch_order = [&](int slot_rounds, int theta) {
        StageTimer t(s, IBGS_STAGE_TILE_ORDER);
        hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, nt, nslots, slot_rounds, im.tile_walked, im.meta, im.tile_order, a.tile_order_out, theta);
        p.order = im.tile_order;
    };
    int risk_slots = 0;          // the flagged tiles' kernel runs over the slots of the launch order when there is one, else over the tiles
    if (a.render_geo) {
        // geo: one wave per tile on large frames (measured at C3-geo: 1.61 ms against 1.83 ms with one wave per half tile), one per
        // 8x8 quadrant on small ones
        const bool big = (a.flags & IBGS_FLAG_QUADRANT_WAVES) ? false : ((a.flags & IBGS_FLAG_TILE_WAVES) ? true : nt >= 4096);
        // The window pass (B2) only carries dL/dmedian_depth and dL/dwarped_image: when neither came in (a loss on `render` / `rendered_normal` alone, e.g. the
        // 2 x len(cameras) iterations in which train.py renders geo without its geo losses, train.py:289-316) every table entry would be zero -- no table is
        // built (0.19 of a 1.90 ms iteration at C3) and the blend loop looks nothing up: bit-identical to zero-filled gradients (tests/test_gpu_geo_no_window.py)
        const bool window = a.dL_ddepth || a.dL_dwarped;
        if (window && !geo_tab) { set_error("geo backward needs the window table scratch"); return -IBGS_ERR_INVALID; }
        if (window) {
            StageTimer t(s, IBGS_STAGE_GEO_WINDOW);
            hipLaunchKernelGGL(geo_window_kernel, dim3((unsigned)(((size_t)a.W * a.H + 255) / 256)), dim3(256), 0, s, p);
        } else p.tab = nullptr;
        IBGS_HIP(hipGetLastError());
        // the geo kernel holds four waves per SIMD: 4 096 slots for a 1080p frame's 8 160 tiles, so the second half of the launch starts as slots
        // free up -- a queue, which wants the heaviest tiles first (plain descending order; slot_rounds = 4 makes the order kernel choose it).  That
        // gives up the 8 x 4 block map's L2 locality and still wins everywhere: C3-geo 1.497 -> 1.452 ms, trained 0.600 -> 0.570, half of the
        // Gaussians in one blob 1.764 -> 1.464 ms (-17 %)
        if (big) {
            launch_order(4, 0);
            risk_slots = nslots;
            StageTimer t(s, IBGS_STAGE_RENDER_BWD);
            if (noabs) hipLaunchKernelGGL(render_bwd_geo4_noabs_kernel, dim3((unsigned)nslots), dim3(64), 0, s, p);
            else hipLaunchKernelGGL(render_bwd_geo4_kernel, dim3((unsigned)nslots), dim3(64), 0, s, p);
        } else if (!(a.flags & IBGS_FLAG_QUADRANT_WAVES) && nt >= HYBRID_MIN_TILES) {
            // small frames: per tile one wave or four (render_bwd_geo_hybrid_kernel), the tiles' first waves heaviest first (slot_rounds = 4: plain descending order)
            launch_order(4, hybrid_theta());
            IBGS_HIP(hipGetLastError());
            p.slab_ipt = 4; p.hybrid_grid1 = nslots;
            risk_slots = nslots;
            StageTimer t(s, IBGS_STAGE_RENDER_BWD);
            if (noabs) hipLaunchKernelGGL(render_bwd_geo_hybrid_kernel<false>, dim3(4u * (unsigned)nslots), dim3(64), 0, s, p);
            else hipLaunchKernelGGL(render_bwd_geo_hybrid_kernel<true>, dim3(4u * (unsigned)nslots), dim3(64), 0, s, p);
        } else {
            StageTimer t(s, IBGS_STAGE_RENDER_BWD);
            hipLaunchKernelGGL(render_bwd_geo_kernel, grid(4), dim3(64), 0, s, p);
        }
    } else if ((a.flags & IBGS_FLAG_QUADRANT_WAVES) ? true : ((a.flags & IBGS_FLAG_TILE_WAVES) ? false : nt < hybrid_max_tiles())) {
        if (!(a.flags & IBGS_FLAG_QUADRANT_WAVES) && nt >= HYBRID_MIN_TILES) {
            // per tile one wave or four (render_bwd_color_hybrid_kernel), the tiles' first waves in the balanced order of the tile-wave kernel below
            launch_order(ORDER_SNAKE_ROUNDS, hybrid_theta());
            IBGS_HIP(hipGetLastError());
            p.slab_ipt = 4; p.hybrid_grid1 = nslots;
            risk_slots = nslots;
            StageTimer t(s, IBGS_STAGE_RENDER_BWD);
            if (noabs) hipLaunchKernelGGL(render_bwd_color_hybrid_kernel<false>, dim3(4u * (unsigned)nslots), dim3(64), 0, s, p);
            else hipLaunchKernelGGL(render_bwd_color_hybrid_kernel<true>, dim3(4u * (unsigned)nslots), dim3(64), 0, s, p);
        } else {
            StageTimer t(s, IBGS_STAGE_RENDER_BWD);
            hipLaunchKernelGGL(render_bwd_color_small_kernel, grid(4), dim3(64), 0, s, p);
        }
    } else {
        // one wave per tile in the balanced (snake) order; docs/EXPERIMENTS.md keeps the numbers of the tile map's order and of fewer waves per SIMD (round 3)
        launch_order(ORDER_SNAKE_ROUNDS, 0);
        IBGS_HIP(hipGetLastError());
        risk_slots = nslots;
        StageTimer t(s, IBGS_STAGE_RENDER_BWD);
        if (noabs) hipLaunchKernelGGL(render_bwd_color_noabs_kernel, dim3((unsigned)nslots), dim3(64), 0, s, p);
        else hipLaunchKernelGGL(render_bwd_color_kernel, dim3((unsigned)nslots), dim3(64), 0, s, p);
    }
    IBGS_HIP(hipGetLastError());
    launch_risk(risk_slots);
    IBGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace ibgs

#ifdef IBGS_COUNT_LANES
extern "C" int ibgs_debug_lanes_bwd(unsigned long long* dst, int reset)
{
    int rc = (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(ibgs::g_lanes_bwd), sizeof(unsigned long long) * 4, 0, hipMemcpyDeviceToHost);
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(ibgs::g_lanes_bwd), z, sizeof(z), 0, hipMemcpyHostToDevice); }
    return rc;
}
#endif
#ifdef IBGS_TRACE_WAVES
extern "C" int ibgs_debug_trace_bwd(void* dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(ibgs::g_trace_bwd), bytes < sizeof(uint4) * ibgs::IBGS_TRACE_MAX ? bytes : sizeof(uint4) * ibgs::IBGS_TRACE_MAX, 0, hipMemcpyDeviceToHost);
}
#endif
