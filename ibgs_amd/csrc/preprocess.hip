// A1 / V1: per-Gaussian projection, EWA covariance, SH -> RGB, tile rectangle.
//
// Behaviour follows the reference's preprocessCUDA / computeCov3D / computeCov2D /
// computeColorFromSH / in_frustum / getRect (DPR/cuda_rasterizer/forward.cu:58-295,
// auxiliary.h:45-60, 143-168) but is organised for gfx950: the per-Gaussian results that the
// blend kernels need are packed into ONE 64-byte record (struct-of-quads) so that the render
// kernels stage a Gaussian with 16-byte loads from a single cache line, instead of the reference's
// five separate arrays (means2D, conic_opacity, rgb, all_map, depths).
//
// This translation unit is compiled with -ffp-contract=off and uses IEEE divide / sqrt so that
// radii, tile rectangles and the record values are bit-identical to the scalar C oracle
// (oracle/ibgs_oracle.c: orc_preprocess) -- the integer outputs (radii, tiles touched) are
// parity-tested exactly.
#include "common.h"
#include <stdlib.h>

namespace ibgs {

__constant__ float kC0 = 0.28209479177387814f;
__constant__ float kC1 = 0.4886025119029199f;
__constant__ float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                             -1.0925484305920792f, 0.5462742152960396f};
__constant__ float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                             0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                             -0.5900435899266435f};

struct PreParams {
    int P, D, M;
    const float* means3D; const float* scales; const float* rotations; const float* opacities;
    const float* shs; const float* shs_rest;          // shs_rest != nullptr: shs holds the DC coefficient (P x 1 x 3), shs_rest the other M - 1 (P x (M - 1) x 3)
    const float* cov3D_precomp; const float* colors_precomp; const float* all_map;
    const float* plane_normal; const float* plane_offset; int plane_mode;
    int inst0;          // batched views: index of this view's first instance in the per-instance outputs (view * P)
    int tile_row0;      // ... and its first row in the stacked tile grid (view * ceil(H/16))
    float scale_modifier;
    int depth_only;
    int32_t* radii;
    float* rec; float* depths; float* cov3D; uint32_t* tiles; uint4* fp; uint64_t* tmask_hi; uint8_t* clamped;
    uint64_t* alive64;      // split mode only (else nullptr): per wave of 64 Gaussians, who reaches a tile list
    uint32_t* sort_key; uint32_t* sort_val;
    int cull;
    uint32_t* zero_a; uint32_t zero_a_n; uint32_t* zero_b; uint32_t zero_b_n;      // words the next stages want zeroed (the depth sort's scratch, its counters)
    RenderedNote note;          // sh_color_kernel: note.host != nullptr -> workgroup 0 adds the tile sums up for the host first (common.h)
    uint32_t* tile_partial; int partial0; int partial_err;      // tiles touched per wave: this launch's first word; the word that follows ALL waves' words (the depth sort's error flag, zeroed here)
};

__device__ __forceinline__ float ndc_to_pix(float v, int S)
{   // auxiliary.h:45-48: evaluated in double because of the double literals
    return (float)((((double)v + 1.0) * S - 1.0) * 0.5);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(hi, max(lo, v)); }

// ---- tile culling ------------------------------------------------------------------------------
// A Gaussian passes the blend's alpha >= 1/255 test only where q = a dx^2 + 2b dx dy + c dy^2 <=
// 2 ln(255 o).  Tiles of the reference rectangle whose pixel-centre box lies outside that ellipse
// (0.1 % + 1e-3 margin, orders of magnitude above the fp32 rounding of `power` in the blend) are
// never emitted: shorter lists, identical images and gradients.  Same arithmetic as
// oracle/ibgs_oracle.c:tile_cull (basic IEEE ops only, so the masks agree bit for bit); the row-wise
// test itself lives in common.h (cull_rows_setup / cull_row_run) because the binning recomputes it for
// rectangles that are too large for a mask.
struct CullJob { CullRows rows; int y0, w, h; };
enum { CULL_NONE = 0, CULL_AABB = 1, CULL_ROWS = 2 };

// Tightens the rectangle to the ellipse's extent and decides how its tiles are tested: CULL_NONE (nothing can pass: no tiles),
// CULL_AABB (a degenerate / near-singular conic: keep the whole rectangle), CULL_ROWS (the tile rows are walked: by the owning lane when
// there are few, by the whole wave when there are many; rectangles of more than IBGS_CULL_MAX_TILES tiles only count their tiles
// here -- mask words all zero -- and the binning recomputes the runs).
__device__ __forceinline__ int cull_setup(float px, float py, float sxx, float syy, float A, float B, float C, float o,
                                          int& x0, int& y0, int& x1, int& y1, CullJob& j)
{
    if (conic_is_risky(A, B, C)) return CULL_AABB;          // near-singular conic: the margin below is not sized for its rounding (common.h); keep the reference rectangle
    const float x255 = 255.0f * o;
    if (!(x255 >= 1.0f)) { x1 = x0; y1 = y0; return CULL_NONE; }
    const float qmax = cull_qmax(o);
    const float hx = sqrtf(qmax * sxx), hy = sqrtf(qmax * syy);
    int tx0 = (int)ceilf((px - hx - 15.0f) / 16.0f), tx1 = (int)floorf((px + hx) / 16.0f) + 1;
    int ty0 = (int)ceilf((py - hy - 15.0f) / 16.0f), ty1 = (int)floorf((py + hy) / 16.0f) + 1;
    tx0 = max(tx0, x0); tx1 = min(tx1, x1); ty0 = max(ty0, y0); ty1 = min(ty1, y1);
    if (tx1 <= tx0 || ty1 <= ty0) { x1 = x0; y1 = y0; return CULL_NONE; }
    x0 = tx0; x1 = tx1; y0 = ty0; y1 = ty1;
    const float det = A * C - B * B;
    if (!(A > 0.0f) || !(C > 0.0f) || !(det > 0.0f)) return CULL_AABB;
    cull_rows_setup(j.rows, px, py, A, B, C, det, qmax, tx0, tx1);
    j.y0 = ty0; j.w = tx1 - tx0; j.h = ty1 - ty0;
    return CULL_ROWS;
}

// bits [start, start + len) of a 256-bit mask (len >= 1)
__device__ __forceinline__ void set_run(uint64_t (&m)[IBGS_CULL_WORDS], int start, int len)
{
#pragma unroll
    for (int k = 0; k < IBGS_CULL_WORDS; k++) {
        const int lo = max(start, 64 * k), hi = min(start + len, 64 * k + 64);
        if (lo < hi) m[k] |= ((hi - lo == 64) ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << (lo - 64 * k);
    }
}

// ---- the row walks of one wave's 64 Gaussians (round 6: flattened) -----------------------------------------------------------------------------------
// Until round 5 the owning lane walked the rows of its rectangle (~110 instructions per row with the IEEE square roots): a wave ran as long as its
// TALLEST rectangle, and on a trained scene (plane-like Gaussians, log-normal sizes) that is 16 rows where the mean is 3.  Now every lane with a job parks
// it in the wave's LDS stage (the record transpose has not started yet) and the wave walks ITEMS = (Gaussian, row) pairs, an item per lane, 64 at a time:
// the owner of an item by binary search in the scanned row counts (six ds_bpermute), the job by a broadcast LDS read, the same cull_row_run on the same
// numbers (common.h: it does not know who calls it), the run OR-ed into the owner's mask words in LDS.  Rectangles of more than CULL_COOP_ROWS rows are
// still taken one at a time, a row per lane, and reduced in registers (their runs span more words than an LDS atomic per item is worth).
// The stage row of a lane (5 quads = 20 words): [0..7] px py B det invA aq ymax ystar, [8] x0 | x1 << 16, [9] y0 | h << 16, [10] tile count, [12..19] mask.
constexpr int CULL_COOP_ROWS = 16;          // rectangles taller than this are walked by the whole wave, one row per lane
__device__ __forceinline__ void park_job(float4* stage_row, const CullJob& j)
{
    stage_row[0] = make_float4(j.rows.px, j.rows.py, j.rows.B, j.rows.det);
    stage_row[1] = make_float4(j.rows.invA, j.rows.aq, j.rows.ymax, j.rows.ystar);
    stage_row[2] = make_float4(__uint_as_float((uint32_t)j.rows.x0 | ((uint32_t)j.rows.x1 << 16)), __uint_as_float((uint32_t)j.y0 | ((uint32_t)j.h << 16)), 0.f, 0.f);
    stage_row[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    stage_row[4] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void fetch_job(const float4* stage_row, CullJob& j)
{
    const float4 q0 = stage_row[0], q1 = stage_row[1], q2 = stage_row[2];
    j.rows.px = q0.x; j.rows.py = q0.y; j.rows.B = q0.z; j.rows.det = q0.w;
    j.rows.invA = q1.x; j.rows.aq = q1.y; j.rows.ymax = q1.z; j.rows.ystar = q1.w;
    const uint32_t xs = __float_as_uint(q2.x), ys = __float_as_uint(q2.y);
    j.rows.x0 = (int)(xs & 0xFFFFu); j.rows.x1 = (int)(xs >> 16);
    j.y0 = (int)(ys & 0xFFFFu); j.h = (int)(ys >> 16); j.w = j.rows.x1 - j.rows.x0;
}
// rows: this lane's row count when its rectangle goes through the flat walk (CULL_ROWS, at most CULL_COOP_ROWS rows), else 0.  The whole wave calls.
__device__ __forceinline__ void wave_flat_rows(float4* stage, int rows, int lane)
{
    uint32_t* words = reinterpret_cast<uint32_t*>(stage);
    int incl = rows;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    const int total = __builtin_amdgcn_readlane(incl, 63);
    for (int base = 0; base < total; base += 64) {          // wave-uniform
        const int item = base + lane;
        int lo = 0;          // number of lanes whose inclusive count is <= item = the owner (incl is non-decreasing)
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) { const int v = __shfl(incl, lo + step - 1, 64); if (v <= item) lo += step; }
        const int owner = min(lo, 63);
        const int r = item - (__shfl(incl, owner, 64) - __shfl(rows, owner, 64));
        if (item < total) {
            CullJob u;
            fetch_job(stage + owner * 5, u);
            int t0, t1;
            if (cull_row_run(u.rows, u.y0 + r, t0, t1)) {
                uint32_t* ow = words + owner * 20;
                atomicAdd(ow + 10, (uint32_t)(t1 - t0 + 1));
                if (u.w * u.h <= IBGS_CULL_MAX_TILES) {
                    const int s0 = r * u.w + (t0 - u.rows.x0), e0 = s0 + (t1 - t0 + 1);          // bits [s0, e0) of the 256-bit mask
                    for (int k = s0 >> 5; k <= (e0 - 1) >> 5; k++) {
                        const int lo_b = max(s0, 32 * k) - 32 * k, hi_b = min(e0, 32 * k + 32) - 32 * k;
                        atomicOr(ow + 12 + k, ((hi_b - lo_b == 32) ? ~0u : ((1u << (hi_b - lo_b)) - 1u)) << lo_b);
                    }
                }
            }
        }
    }
}

// One thread per Gaussian. The AoS inputs (12..192 B per Gaussian) are read with plain per-lane
// loads; a wave touches a contiguous span of each array, so every fetched line is fully used.
// WITH_SH = false: the SH -> RGB evaluation is left to sh_color_kernel below (the launcher's default when SH coefficients are given):
// without the 48 coefficients and 16 basis values live next to the cull state this kernel needs far fewer registers.
template <bool WITH_SH>
__global__ void __launch_bounds__(256, 4) preprocess_kernel(PreParams p, Cam cam)
{
    __shared__ float4 s_stage_all[4][64 * 5];                 // 64 records at a stride of 5 quads (conflict-free 16-byte LDS accesses)
    float4* s_stage = s_stage_all[threadIdx.x >> 6];          // private to the wave: LDS operations of one wave execute in order, no barrier needed
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t z = (uint32_t)gi; z < p.zero_a_n; z += gridDim.x * blockDim.x) p.zero_a[z] = 0u;      // instead of two fill launches
    if ((uint32_t)gi < p.zero_b_n) p.zero_b[gi] = 0u;
    const bool valid = gi < p.P;          // lanes past the end stay in the wave: the record transpose below is wave-wide
    const int i = valid ? gi : p.P - 1;   // (they recompute the last Gaussian and store nothing)

    // defaults: culled Gaussians keep radius 0, zero tiles and sort last
    float rec[REC_FLOATS];
#pragma unroll
    for (int k = 0; k < REC_FLOATS; k++) rec[k] = 0.f;
    int radius = 0; uint32_t ntiles = 0; float depth = 0.f; uint8_t clampbits = 0;
    float c6loc[6] = {0, 0, 0, 0, 0, 0};

    const float px3 = p.means3D[3 * i], py3 = p.means3D[3 * i + 1], pz3 = p.means3D[3 * i + 2];
    const float* __restrict__ vm = cam.vm; const float* __restrict__ pm = cam.pm;
    const float hx = pm[0] * px3 + pm[4] * py3 + pm[8] * pz3 + pm[12];
    const float hy = pm[1] * px3 + pm[5] * py3 + pm[9] * pz3 + pm[13];
    const float hw = pm[3] * px3 + pm[7] * py3 + pm[11] * pz3 + pm[15];
    const float pw = 1.0f / (hw + 0.0000001f);
    const float zview = vm[2] * px3 + vm[6] * py3 + vm[10] * pz3 + vm[14];

    bool alive = zview > 0.2f;   // !(z <= 0.2) differs only for NaN, which is culled either way
    // (values of the first half -- up to the cull decision -- that the second half needs; the wave walks its rectangles' rows in between, all lanes together)
    float ca = 0.f, cb = 0.f, cc = 0.f, det_inv = 0.f, pxs = 0.f, pys = 0.f;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    int walk_rows = 0;          // rows of this lane's rectangle that the wave has to walk (0: no row cull)
    bool full_mask = false;     // every tile of the rectangle (no cull, or a conic the cull leaves alone)
    const int lane = threadIdx.x & 63;
    if (alive) {
        // ---- 3D covariance (upper triangle) ----
        if (p.cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) c6loc[k] = p.cov3D_precomp[6 * i + k];
        } else {
            const float r = p.rotations[4 * i], x = p.rotations[4 * i + 1], y = p.rotations[4 * i + 2], z = p.rotations[4 * i + 3];
            float R[3][3];
            R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z);       R[0][2] = 2.f * (x * z + r * y);
            R[1][0] = 2.f * (x * y + r * z);       R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
            R[2][0] = 2.f * (x * z - r * y);       R[2][1] = 2.f * (y * z + r * x);       R[2][2] = 1.f - 2.f * (x * x + y * y);
            const float sv[3] = {p.scale_modifier * p.scales[3 * i], p.scale_modifier * p.scales[3 * i + 1],
                                 p.scale_modifier * p.scales[3 * i + 2]};
            float Mm[3][3];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) Mm[a][b] = sv[a] * R[b][a];
#define SIG(a, b) (Mm[0][a] * Mm[0][b] + Mm[1][a] * Mm[1][b] + Mm[2][a] * Mm[2][b])
            c6loc[0] = SIG(0, 0); c6loc[1] = SIG(0, 1); c6loc[2] = SIG(0, 2);
            c6loc[3] = SIG(1, 1); c6loc[4] = SIG(1, 2); c6loc[5] = SIG(2, 2);
#undef SIG
        }
        // ---- EWA 2D covariance ----
        float t0 = vm[0] * px3 + vm[4] * py3 + vm[8] * pz3 + vm[12];
        float t1 = vm[1] * px3 + vm[5] * py3 + vm[9] * pz3 + vm[13];
        const float t2 = vm[2] * px3 + vm[6] * py3 + vm[10] * pz3 + vm[14];
        const float limx = 1.3f * cam.tanfovx, limy = 1.3f * cam.tanfovy;
        const float txtz = t0 / t2, tytz = t1 / t2;
        t0 = fminf(limx, fmaxf(-limx, txtz)) * t2;
        t1 = fminf(limy, fmaxf(-limy, tytz)) * t2;
        const float j00 = cam.fx / t2, j02 = -(cam.fx * t0) / (t2 * t2);
        const float j11 = cam.fy / t2, j12 = -(cam.fy * t1) / (t2 * t2);
        float A[2][3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const float rv0 = vm[4 * r + 0], rv1 = vm[4 * r + 1], rv2 = vm[4 * r + 2];
            A[0][r] = rv0 * j00 + rv1 * 0.0f + rv2 * j02;
            A[1][r] = rv0 * 0.0f + rv1 * j11 + rv2 * j12;
        }
        const float S[3][3] = {{c6loc[0], c6loc[1], c6loc[2]}, {c6loc[1], c6loc[3], c6loc[4]}, {c6loc[2], c6loc[4], c6loc[5]}};
        float SA[2][3];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int r = 0; r < 3; r++) SA[a][r] = S[r][0] * A[a][0] + S[r][1] * A[a][1] + S[r][2] * A[a][2];
        ca = A[0][0] * SA[0][0] + A[0][1] * SA[0][1] + A[0][2] * SA[0][2] + 0.3f;
        cb = A[0][0] * SA[1][0] + A[0][1] * SA[1][1] + A[0][2] * SA[1][2];
        cc = A[1][0] * SA[1][0] + A[1][1] * SA[1][1] + A[1][2] * SA[1][2] + 0.3f;
        const float det = ca * cc - cb * cb;
        alive = (det != 0.0f);
        if (alive) {
            det_inv = 1.f / det;
            const float mid = 0.5f * (ca + cc);
            const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lam1 = mid + disc, lam2 = mid - disc;
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lam1, lam2)));
            pxs = ndc_to_pix(hx * pw, cam.W); pys = ndc_to_pix(hy * pw, cam.H);
            const int rad = (int)my_radius;
            x0 = clampi((int)((pxs - rad) / TILE), 0, cam.gx);
            y0 = clampi((int)((pys - rad) / TILE), 0, cam.gy);
            x1 = clampi((int)((pxs + rad + TILE - 1) / TILE), 0, cam.gx);
            y1 = clampi((int)((pys + rad + TILE - 1) / TILE), 0, cam.gy);
            alive = ((x1 - x0) * (y1 - y0)) != 0;
            if (alive) {
                radius = rad; ntiles = (uint32_t)((x1 - x0) * (y1 - y0));
                full_mask = true;
                if (p.cull) {
                    CullJob job;
                    const int cull_mode = cull_setup(pxs, pys, ca, cc, cc * det_inv, -cb * det_inv, ca * det_inv, p.opacities[i], x0, y0, x1, y1, job);
                    if (cull_mode == CULL_NONE) { ntiles = 0; full_mask = false; }
                    else if (cull_mode == CULL_AABB) ntiles = (uint32_t)((x1 - x0) * (y1 - y0));
                    else {          // CULL_ROWS: the wave walks the rows, right below
                        full_mask = false;
                        walk_rows = job.h;
                        park_job(s_stage + lane * 5, job);
                    }
                }
            }
        }
    }
    // ---- the row walks, all lanes together (see wave_flat_rows above); the stage is not yet in use for the records ----
    if (p.cull) {          // uniform
        const bool tall = walk_rows > CULL_COOP_ROWS;
        if (__ballot(walk_rows != 0) != 0ull) {          // wave-uniform
            wave_flat_rows(s_stage, tall ? 0 : walk_rows, lane);
            uint64_t todo = __ballot(tall);
            while (todo != 0ull) {          // wave-uniform: the heavy tail of a trained scene, a per cent or two of its Gaussians -- one at a time, a row per lane
                const int g = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                CullJob u;
                fetch_job(s_stage + g * 5, u);
                const bool masked = u.w * u.h <= IBGS_CULL_MAX_TILES;
                uint64_t m[IBGS_CULL_WORDS] = {0, 0, 0, 0};
                uint32_t cnt = 0;
                for (int r = lane; r < u.h; r += 64) {
                    int t0, t1;
                    if (!cull_row_run(u.rows, u.y0 + r, t0, t1)) continue;
                    if (masked) set_run(m, r * u.w + (t0 - u.rows.x0), t1 - t0 + 1);
                    cnt += (uint32_t)(t1 - t0 + 1);
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, d, 64);
                uint32_t* ow = reinterpret_cast<uint32_t*>(s_stage) + g * 20;          // into the owner's stage row, like the flat walk's results
                if (masked) {          // wave-uniform
#pragma unroll
                    for (int k = 0; k < IBGS_CULL_WORDS; k++) {
                        uint32_t lo = (uint32_t)m[k], hi = (uint32_t)(m[k] >> 32);
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) { lo |= (uint32_t)__shfl_xor((int)lo, d, 64); hi |= (uint32_t)__shfl_xor((int)hi, d, 64); }
                        if (lane == 0) { ow[12 + 2 * k] = lo; ow[13 + 2 * k] = hi; }
                    }
                }
                if (lane == 0) ow[10] = cnt;
            }
        }
    }
    {   // the footprint leaves right here (rectangle + mask word 0: one 16-byte record; words 1..3 only matter -- and are only read by the binning -- for
        // rectangles of more than 64 tiles), so that the mask is not carried through the second half
        uint64_t tmask[IBGS_CULL_WORDS];
        const uint32_t* ow = reinterpret_cast<const uint32_t*>(s_stage) + lane * 20;
#pragma unroll
        for (int k = 0; k < IBGS_CULL_WORDS; k++) tmask[k] = walk_rows != 0 ? (((uint64_t)ow[13 + 2 * k] << 32) | ow[12 + 2 * k]) : (full_mask ? ~0ull : 0ull);
        if (walk_rows != 0) ntiles = ow[10];
        uint32_t rx = 0, ry = 0;
        bool big_rect = false;
        if (alive) {
            big_rect = (x1 - x0) * (y1 - y0) > 64 && (x1 - x0) * (y1 - y0) <= IBGS_CULL_MAX_TILES;          // mask words 1..3 in use
            rx = pack_rect(x0, x1); ry = pack_rect(y0 + (uint32_t)p.tile_row0, y1 + (uint32_t)p.tile_row0);
        }
        if (valid) {
            const int o = p.inst0 + i;            // instance slot (= i for a single view)
            p.fp[o] = make_uint4(rx, ry, (uint32_t)tmask[0], (uint32_t)(tmask[0] >> 32));
            if (big_rect) {
#pragma unroll
                for (int k = 1; k < IBGS_CULL_WORDS; k++) p.tmask_hi[(size_t)o * (IBGS_CULL_WORDS - 1) + (k - 1)] = tmask[k];
            }
        }
    }
    if (alive) {          // ---- second half: the record ----
        depth = zview;
        rec[R_X] = pxs; rec[R_Y] = pys; rec[R_OP] = p.opacities[i];
        rec[R_CA] = cc * det_inv; rec[R_CB] = -cb * det_inv; rec[R_CC] = ca * det_inv;
        if (p.colors_precomp) {
            rec[R_R] = p.colors_precomp[3 * i]; rec[R_G] = p.colors_precomp[3 * i + 1]; rec[R_B] = p.colors_precomp[3 * i + 2];
        } else if (WITH_SH && !p.depth_only) {
            // ---- SH -> RGB ----
            float d0 = px3 - cam.campos[0], d1 = py3 - cam.campos[1], d2 = pz3 - cam.campos[2];
            const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
            d0 /= len; d1 /= len; d2 /= len;
            float B[16];
            int nb = 1;
            B[0] = kC0;
            if (p.D > 0) {
                const float x = d0, y = d1, z = d2;
                B[1] = -kC1 * y; B[2] = kC1 * z; B[3] = -kC1 * x; nb = 4;
                if (p.D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    B[4] = kC2[0] * xy; B[5] = kC2[1] * yz; B[6] = kC2[2] * (2.0f * zz - xx - yy);
                    B[7] = kC2[3] * xz; B[8] = kC2[4] * (xx - yy); nb = 9;
                    if (p.D > 2) {
                        B[9] = kC3[0] * y * (3.0f * xx - yy);
                        B[10] = kC3[1] * xy * z;
                        B[11] = kC3[2] * y * (4.0f * zz - xx - yy);
                        B[12] = kC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                        B[13] = kC3[4] * x * (4.0f * zz - xx - yy);
                        B[14] = kC3[5] * z * (xx - yy);
                        B[15] = kC3[6] * x * (xx - 3.0f * yy);
                        nb = 16;
                    }
                }
            }
            // All coefficient loads are issued back to back (compile-time unrolled, 16-B vector loads
            // when rows are 16-B aligned) so that they overlap instead of one dependent wait per term.
            float shv[48];
            const float* sh = p.shs + (size_t)i * p.M * 3;
            const int need = 3 * nb;
            if (p.shs_rest) {          // DC and the rest in two arrays (the model's f_dc / f_rest as they are): plain per-lane loads
                const float* sr = p.shs_rest + (size_t)i * (p.M - 1) * 3;
                const float* sd = p.shs + (size_t)i * 3;
#pragma unroll
                for (int k = 0; k < 48; k++) if (k < need) shv[k] = k < 3 ? sd[k] : sr[k - 3];
            } else if (p.M == 16) {
                const float4* r4 = reinterpret_cast<const float4*>(sh);
#pragma unroll
                for (int v = 0; v < 12; v++) {
                    if (4 * v < need) {
                        const float4 q = r4[v];
                        shv[4 * v] = q.x; shv[4 * v + 1] = q.y; shv[4 * v + 2] = q.z; shv[4 * v + 3] = q.w;
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 48; k++) if (k < need) shv[k] = sh[k];
            }
            float col[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) col[ch] = B[0] * shv[ch];
#pragma unroll
            for (int k = 1; k < 16; k++) {
                if (k < nb) {
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) col[ch] = col[ch] + B[k] * shv[3 * k + ch];
                }
            }
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float v = col[ch] + 0.5f;
                if (v < 0) clampbits |= (uint8_t)(1u << ch);
                rec[R_R + ch] = fmaxf(v, 0.0f);
            }
        }
        if (p.plane_mode) {
            const PlaneEval e = plane_eval(p.plane_mode, p.plane_normal, p.plane_offset, p.scales, p.rotations, i, px3, py3, pz3, cam.campos, vm);
            rec[R_NX] = e.ncam[0]; rec[R_NY] = e.ncam[1]; rec[R_NZ] = e.ncam[2]; rec[R_DIST] = e.dist;
        } else if (p.all_map) {
            rec[R_NX] = p.all_map[5 * i]; rec[R_NY] = p.all_map[5 * i + 1]; rec[R_NZ] = p.all_map[5 * i + 2];
            rec[R_DIST] = p.all_map[5 * i + 4];
        }
    }
    if (!alive) {
#pragma unroll
        for (int k = 0; k < REC_FLOATS; k++) rec[k] = 0.f;
    }
    {   // the 64-byte records leave through LDS: a lane storing its own record spreads every store instruction over 64 lines; transposed, the
        // wave writes its 4 KB as four fully coalesced 1 KB stores
#pragma unroll
        for (int k = 0; k < 4; k++) s_stage[lane * 5 + k] = make_float4(rec[4 * k], rec[4 * k + 1], rec[4 * k + 2], rec[4 * k + 3]);
        const int first = blockIdx.x * blockDim.x + (threadIdx.x & ~63);          // this wave's Gaussians are [first, first + 64) cut at P
        const int nrow = min(64, p.P - first);
        float4* out = reinterpret_cast<float4*>(p.rec) + ((size_t)p.inst0 + first) * 4;
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int q = it * 64 + lane, row = q >> 2;
            if (row < nrow) out[q] = s_stage[row * 5 + (q & 3)];
        }
    }
    if (p.alive64) {          // (a full-wave ballot: lanes past the end are still here)
        const uint64_t am = __ballot(valid && alive && ntiles > 0);
        if ((threadIdx.x & 63) == 0) p.alive64[gi >> 6] = am;
    }
    {   // tiles touched by this wave's Gaussians: the host adds the waves' words up and has R right after the depth sort (api.hip), without a scan
        uint32_t tsum = valid ? ntiles : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) tsum += (uint32_t)__shfl_xor((int)tsum, d, 64);
        if ((threadIdx.x & 63) == 0) p.tile_partial[p.partial0 + (gi >> 6)] = tsum;
        if (gi == 0) p.tile_partial[p.partial_err] = 0u;
    }
    if (!valid) return;

    const int o = p.inst0 + i;            // instance slot (= i for a single view)
    p.radii[o] = radius;
    p.tiles[o] = ntiles;
    p.depths[o] = depth;
    p.clamped[o] = clampbits;
    if (!p.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) p.cov3D[6 * o + k] = c6loc[k];   // computed for every Gaussian past the near cull
    }
    // depth sort input: positive float bits order like the floats; culled Gaussians sort last
    // Gaussians without any tile (culled, or fully tile-culled) sort last and emit nothing
    p.sort_key[o] = (alive && ntiles > 0) ? __float_as_uint(depth) : 0xFFFFFFFFu;
    p.sort_val[o] = (uint32_t)o;
}

// ---- SH -> RGB as its own pass (forward.cu:58-109, 280-286) ------------------------------------------------------------------------
// One wave per 64 consecutive Gaussians.  Their coefficient rows are one contiguous 12 KB block (M = 16): the wave fetches it with twelve
// fully coalesced 1 KB loads (float4 per lane, all in flight together) -- skipping the 16-byte pieces of rows whose Gaussian reaches no
// tile list (culled, off screen: 40 % of C3; the preprocess kernel left a lane mask per wave) -- and transposes it through LDS in two
// rounds of 32 rows (rows padded to 52 words: conflict-free 16-byte accesses both ways; 6.5 KB per wave, so the register budget and not
// LDS sets the occupancy): round h parks rows 32h .. 32h + 31, lane 32h + r reads row r back.  Every load instruction covers one
// contiguous kilobyte, every cache line is fetched once.  In the step it runs at ~3 TB/s of useful bytes -- what a 100-200 MB read gets
// right after kernels that left the caches full of dirty lines (tests/csrc/probe_read_bw.hip "cold": 2.9 TB/s; 6.2 TB/s when nothing
// has to drain), whatever its occupancy or load shape (three variants measured, docs/EXPERIMENTS.md section 7).  The evaluation is the oracle's, operation by
// operation and in its order (this file is compiled without contraction): colours and clamp flags stay bit-identical.  Writes quad 2 of
// the render record and the clamp bits.
constexpr int SHC_ROW = 52;          // LDS words per row: 48 coefficients + 4 words of padding
template <bool SPLIT>          // SPLIT: DC and rest coefficients in two arrays (ibgs_forward_args.shs_rest) -- its own instantiation, so that the combined layout's code stays what it was
__global__ void __launch_bounds__(256) sh_color_kernel(PreParams p, Cam cam)
{
    __shared__ float s_sh_all[4][32 * SHC_ROW];
    if (p.note.host != nullptr && blockIdx.x == 0) {          // (workgroup-uniform) R for the host, before anything else: it sizes the tile lists with it
        __shared__ unsigned long long s_note[4];
        rendered_note_block<256>(p.note, s_note);
    }
    float* s_sh = s_sh_all[threadIdx.x >> 6];          // private to the wave (LDS operations of one wave execute in order)
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int first = wave * 64;          // this wave's Gaussians: [first, first + 64) cut at P
    if (first >= p.P) return;
    const uint64_t alive_m = p.alive64[wave];          // wave-uniform
    if (alive_m == 0ull) return;
    const int i = min(first + lane, p.P - 1);
    const bool alive = (alive_m >> lane) & 1ull;
    const int D = p.D, M = p.M;
    const int nb = (D + 1) * (D + 1);
    const float px3 = p.means3D[3 * i], py3 = p.means3D[3 * i + 1], pz3 = p.means3D[3 * i + 2];
    float shv[48];
    if (SPLIT && M == 16 && first + 64 <= p.P) {
        // DC and the rest in two arrays (the reference model's `_features_dc` (P, 1, 3) and `_features_rest` (P, 15, 3) as they are: no torch.cat, no
        // second copy of 192 B per Gaussian).  Same scheme as below: the wave's 64 rows are one contiguous block per array -- 64 x 180 B = 720 float4 of
        // rest, 64 x 12 B = 48 float4 of DC, 768 = 12 x 64 pieces, all loads coalesced and in flight together -- parked in LDS in two rounds of 32 rows
        // (360 + 24 = 384 = 6 x 64 pieces per round) AS THEY LIE in memory (16-byte LDS stores, conflict-free); lane r then reads its row word by word:
        // the rows are 45 (and 3) words apart, odd strides, so the 64 lanes of every read hit distinct banks.  (Scattering the pieces into padded rows
        // instead -- word stores 4 apart -- ran into 8-way bank conflicts: sh_color 51 -> 83 us.)  Pieces that hold nothing a live Gaussian needs are skipped.
        const float4* rest4 = reinterpret_cast<const float4*>(p.shs_rest + (size_t)first * 45);
        const float4* dc4 = reinterpret_cast<const float4*>(p.shs + (size_t)first * 3);
        float4 v[12];
#pragma unroll
        for (int it = 0; it < 12; it++) {
            const int h = it / 6, qp = (it % 6) * 64 + lane;          // piece of round h
            if (qp < 360) {
                const int ra = (4 * qp) / 45, rb = (4 * qp + 3) / 45;          // the (at most two) rows the piece touches
                const bool need = nb > 1 && (((alive_m >> (32 * h + ra)) & 1ull) || ((alive_m >> (32 * h + rb)) & 1ull));
                v[it] = need ? rest4[360 * h + qp] : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const int ra = (4 * (qp - 360)) / 3, rb = (4 * (qp - 360) + 3) / 3;
                const bool need = (((alive_m >> (32 * h + ra)) & 3ull) != 0ull) || ((alive_m >> (32 * h + rb)) & 1ull);          // (a DC piece touches rows ra, ra + 1 [, rb])
                v[it] = need ? dc4[24 * h + (qp - 360)] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float4* s_raw = reinterpret_cast<float4*>(s_sh);          // 384 float4 of this round: rest rows 0..31 (1440 words), then their DC (96 words)
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int it = 0; it < 6; it++) s_raw[it * 64 + lane] = v[h * 6 + it];
            if ((lane >> 5) == h) {
                const float* rr = s_sh + 45 * (lane & 31);
                const float* rd = s_sh + 1440 + 3 * (lane & 31);
                shv[0] = rd[0]; shv[1] = rd[1]; shv[2] = rd[2];
#pragma unroll
                for (int k = 0; k < 45; k++) shv[3 + k] = rr[k];
            }
        }
    } else if (SPLIT) {          // other coefficient counts / the last, partial wave: plain per-lane loads from the two arrays
        const float* sr = p.shs_rest + (size_t)i * (M - 1) * 3;
        const float* sd = p.shs + (size_t)i * 3;
#pragma unroll
        for (int k = 0; k < 48; k++) shv[k] = (alive && k < 3 * nb) ? (k < 3 ? sd[k] : sr[k - 3]) : 0.f;
    } else if (M == 16) {
        const float4* src = reinterpret_cast<const float4*>(p.shs) + (size_t)first * 12;
        const int nq = min(64, p.P - first) * 12;          // float4 pieces of this wave's block
        float4 v[12];
#pragma unroll
        for (int it = 0; it < 12; it++) {
            const int q = it * 64 + lane;
            const int row = q / 12, piece = q - row * 12;
            const bool ok = q < nq && ((alive_m >> row) & 1ull) && 4 * piece < 3 * nb;
            v[it] = ok ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const int q = it * 64 + lane;          // piece index within this round's 32 rows
                const int row = q / 12, piece = q - row * 12;
                *reinterpret_cast<float4*>(s_sh + row * SHC_ROW + piece * 4) = v[h * 6 + it];
            }
            if ((lane >> 5) == h) {
                const float4* row4 = reinterpret_cast<const float4*>(s_sh + (lane & 31) * SHC_ROW);
#pragma unroll
                for (int k = 0; k < 12; k++) { const float4 q = row4[k]; shv[4 * k] = q.x; shv[4 * k + 1] = q.y; shv[4 * k + 2] = q.z; shv[4 * k + 3] = q.w; }
            }
        }
    } else {          // other coefficient counts: plain per-lane loads of the row's first 3 * nb floats
        const float* sh = p.shs + (size_t)i * M * 3;
#pragma unroll
        for (int k = 0; k < 48; k++) shv[k] = (alive && k < 3 * nb) ? sh[k] : 0.f;
    }
    if (!alive) return;
    float d0 = px3 - cam.campos[0], d1 = py3 - cam.campos[1], d2 = pz3 - cam.campos[2];
    const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    d0 /= len; d1 /= len; d2 /= len;
    float B[16];
    B[0] = kC0;
    if (D > 0) {
        const float x = d0, y = d1, z = d2;
        B[1] = -kC1 * y; B[2] = kC1 * z; B[3] = -kC1 * x;
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = kC2[0] * xy; B[5] = kC2[1] * yz; B[6] = kC2[2] * (2.0f * zz - xx - yy);
            B[7] = kC2[3] * xz; B[8] = kC2[4] * (xx - yy);
            if (D > 2) {
                B[9] = kC3[0] * y * (3.0f * xx - yy);
                B[10] = kC3[1] * xy * z;
                B[11] = kC3[2] * y * (4.0f * zz - xx - yy);
                B[12] = kC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                B[13] = kC3[4] * x * (4.0f * zz - xx - yy);
                B[14] = kC3[5] * z * (xx - yy);
                B[15] = kC3[6] * x * (xx - 3.0f * yy);
            }
        }
    }
    float col[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) col[ch] = B[0] * shv[ch];
#pragma unroll
    for (int k = 1; k < 16; k++) {
        if (k < nb) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) col[ch] = col[ch] + B[k] * shv[3 * k + ch];
        }
    }
    const int o = p.inst0 + i;
    uint8_t clampbits = 0;
    float4 out;
    float* oc = &out.x;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float v = col[ch] + 0.5f;
        if (v < 0) clampbits |= (uint8_t)(1u << ch);
        oc[ch] = fmaxf(v, 0.0f);
    }
    out.w = 0.f;
    reinterpret_cast<float4*>(p.rec)[(size_t)o * 4 + 2] = out;
    p.clamped[o] = clampbits;
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* means3D, Cam cam, uint8_t* present)
{   // checkFrustum, rasterizer_impl.cu:171-183 (only the z test is live, auxiliary.h:158)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float z = cam.vm[2] * means3D[3 * i] + cam.vm[6] * means3D[3 * i + 1] + cam.vm[10] * means3D[3 * i + 2] + cam.vm[14];
    present[i] = z > 0.2f ? 1 : 0;
}

static bool preprocess_is_split(const ibgs_forward_args& a)          // geometry and SH colours in two kernels
{
    return a.shs && !a.colors_precomp && !a.render_depth_only;
}

int launch_preprocess(hipStream_t s, const ibgs_forward_args& a, const GeomState& g, int phase, const RenderedNote* note)
{   // phase 0: everything; 1: the geometry kernel(s) alone -- after them the tiles-touched sums are final; 2: what phase 1 left out (the SH colours)
    PreParams p;
    p.note = RenderedNote{nullptr, 0u, 0u, nullptr};
    int carried = 0;
    p.P = a.P; p.D = a.D; p.M = a.M;
    p.means3D = a.means3D; p.scales = a.scales; p.rotations = a.rotations; p.opacities = a.opacities;
    p.shs = a.shs; p.shs_rest = a.shs_rest; p.cov3D_precomp = a.cov3D_precomp; p.colors_precomp = a.colors_precomp; p.all_map = a.all_map;
    p.plane_normal = a.plane_normal; p.plane_offset = a.plane_offset; p.plane_mode = a.plane_mode;
    p.scale_modifier = a.scale_modifier; p.depth_only = a.render_depth_only;
    p.radii = a.radii; p.rec = g.rec; p.depths = g.depths; p.cov3D = g.cov3D; p.tiles = g.tiles; p.fp = g.fp; p.tmask_hi = g.tmask_hi;
    p.clamped = g.clamped; p.sort_key = g.sort_key[0]; p.sort_val = g.sort_val[0];
    // depth-only with a 1-slot buffer depends on list positions (the per-round 'break' of forward.cu:484-488)
    p.cull = !(a.flags & IBGS_FLAG_NO_TILE_CULL) && !(a.render_depth_only && a.buffer_length == 1);
    const int blocks = (a.P + 255) / 256;
    const int nv = a.n_views > 1 ? a.n_views : 1;
    const int gy = (a.H + TILE - 1) / TILE;
    p.zero_a = g.hist; p.zero_a_n = (uint32_t)radix_zero_elems((size_t)nv * a.P, 32);
    p.zero_b = g.offsets + (size_t)nv * a.P + 1; p.zero_b_n = 4;          // error flag, C, kept, "the order lies in sort_val[1]"
    for (int v = 0; v < nv; v++) {       // batched depth passes: one launch per camera, outputs land in that view's slice
        p.inst0 = v * a.P; p.tile_row0 = v * gy;
        p.tile_partial = g.tile_partial; p.partial0 = v * ((a.P + 63) / 64); p.partial_err = nv * ((a.P + 63) / 64);
        const Cam cam = make_cam(a.viewmatrix + 16 * v, a.projmatrix + 16 * v, a.campos + 3 * v, a.bg,
                                 nv > 1 ? a.view_tanfovx[v] : a.tanfovx, nv > 1 ? a.view_tanfovy[v] : a.tanfovy, a.W, a.H);
        // SH coefficients: geometry first (few registers, every Gaussian), then the colours of the Gaussians that reach a tile
        // (coalesced row loads)
        const bool split = preprocess_is_split(a);
        p.alive64 = split ? g.alive64 : nullptr;
        // no SH evaluation at all (depth-only passes, precomputed colours): the geometry kernel alone -- the one-kernel form carries the SH path's
        // registers (120 VGPRs, 4 waves per SIMD against 78 / 6 -- 65 / 7 until the row walk was flattened --) whether it runs or not: 73 -> ~50 us per source view of a test-time frame
        const bool no_sh = a.render_depth_only || a.colors_precomp || !a.shs;
        if (split) {
            if (phase != 2) hipLaunchKernelGGL(preprocess_kernel<false>, dim3(blocks), dim3(256), 0, s, p, cam);
            if (phase != 1) {
                if (note && !carried) { p.note = *note; carried = 1; } else p.note.host = nullptr;          // (the first view's launch carries it)
                if (p.shs_rest) hipLaunchKernelGGL(sh_color_kernel<true>, dim3((a.P + 255) / 256), dim3(256), 0, s, p, cam);
                else hipLaunchKernelGGL(sh_color_kernel<false>, dim3((a.P + 255) / 256), dim3(256), 0, s, p, cam);
            }
        } else if (phase != 2) hipLaunchKernelGGL(preprocess_kernel<false>, dim3(blocks), dim3(256), 0, s, p, cam);          // (no_sh: nothing else can get here)
    }
    IBGS_HIP(hipGetLastError());
    return carried;
}

int launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* vm, uint8_t* present)
{
    const Cam cam = make_cam(vm, nullptr, nullptr, nullptr, 1.f, 1.f, 16, 16);
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, cam, present);
    IBGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace ibgs
