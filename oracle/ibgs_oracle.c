/*
 * ibgs_oracle.c -- CPU restatement of the IBGS plane rasterizer (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity oracle for the HIP path in ibgs_amd/csrc. It is plain scalar C,
 * one pixel / one Gaussian at a time, written from the behaviour of the reference CUDA
 * extension (paths below are relative to /root/reference/submodules/diff-plane-rasterization/).
 * It must only ever be used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg -- never by the product path.
 *
 * PARITY STATUS: "parity unpinned" -- the reference ships no tests, golden vectors or
 * known-answer fixtures for this path and its CUDA sources cannot be
 * built here (no nvcc, CUDA textures, CUB). The oracle is pinned only indirectly:
 *   - SH evaluation / camera matrices against values produced by the reference's importable
 *     python helpers (tests/golden/make_golden.py -> tests/golden/ npz files),
 *   - hand-computable micro scenes (tests/test_oracle_kat.py),
 *   - torch.autograd through an independent differentiable restatement of the colour path
 *     (tests/test_oracle_autograd.py) for the backward.
 *
 * Arithmetic: fp32 with IEEE semantics (build with -ffp-contract=off); places where the
 * reference promotes to double through a double literal are done in double here too and
 * are marked "dbl". Gradient accumulation over pixels is done in double (the reference
 * uses float atomics in nondeterministic order; the exact sum is the fair target).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* Per-Gaussian gradient sums over pixels.  Default: double accumulators (see "Arithmetic" above).  -DORC_ACC_FLOAT (oracle.variant("acc32")): every addition is
 * rounded to float, i.e. what the reference's float atomicAdd (backward.cu:673, 770, 793-804) and any fp32 implementation keep -- in ONE thread and pixel order
 * here, so the result is reproducible.  A diagnostic: how far float accumulation alone moves an ill-conditioned gradient (tools/fuzz_parity.py). */
#ifdef ORC_ACC_FLOAT
#define ACC_ADD(dst, v) do { (dst) = (double)(float)((float)(dst) + (float)(v)); } while (0)
#else
#define ACC_ADD(dst, v) do { _Pragma("omp atomic") (dst) += (double)(v); } while (0)
#endif
#include <string.h>

/* Third build of this one source (oracle.variant("f64"), gcc -DORC_F64): every `float` of the restatement -- arrays at the interface
 * included -- becomes a double and every libm call its double form.  Same algorithm, same decisions wherever they do not hinge on the last
 * bit of an fp32 value; the arbiter for ill-conditioned quantities: a float build (the oracle proper, its fma twin, the HIP path) is as
 * good as its distance from this one (tests/test_gpu_anisotropic.py).  The fp32 constants of the source (0.3f, 1.0f / 255 ...) keep
 * their fp32 values. */
typedef float orc_f32;
#ifdef ORC_F64
#define float double
#define sqrtf sqrt
#define expf exp
#define floorf floor
#define ceilf ceil
#define fabsf fabs
#endif

#define TILE 16
#define ROUND 256          /* entries fetched per cooperative round: forward.cu:357,404 */
#define MAX_L 8            /* auxiliary.h:21 */
#define MAX_SRC 5          /* auxiliary.h:22-23 */

/* SH constants: cuda_rasterizer/auxiliary.h:26-43 */
static const float kC0 = 0.28209479177387814f;
static const float kC1 = 0.4886025119029199f;
static const float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                             -1.0925484305920792f, 0.5462742152960396f};
static const float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                             0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                             -0.5900435899266435f};

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:45-48 (dbl: the literals 1.0 / 0.5 are double) */
static inline float ndc_to_pix(float v, int S) { return (float)((((double)v + 1.0) * S - 1.0) * 0.5); }

/* auxiliary.h:50-60. Truncating int conversion, then clamp to the grid. */
static void tile_rect(float px, float py, int radius, int gx, int gy, int* x0, int* y0, int* x1, int* y1)
{
    *x0 = imin(gx, imax(0, (int)((px - radius) / TILE)));
    *y0 = imin(gy, imax(0, (int)((py - radius) / TILE)));
    *x1 = imin(gx, imax(0, (int)((px + radius + TILE - 1) / TILE)));
    *y1 = imin(gy, imax(0, (int)((py + radius + TILE - 1) / TILE)));
}

/* rasterizer_impl.cu:152-167 */
uint32_t orc_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* Standard rotation of a (w,x,y,z) quaternion, NOT normalised (forward.cu:165-176). */
static void quat_to_rot(const float* q, float R[3][3])
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z);       R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z);       R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y);       R[2][1] = 2.f * (y * z + r * x);       R[2][2] = 1.f - 2.f * (x * x + y * y);
}

/* Sigma = Rq diag(mod*s)^2 Rq^T, upper triangle (forward.cu:156-190). M[i][a] = s_i * Rq[a][i]. */
static void cov3d_from_scale_rot(const float* s, float mod, const float* q, float* c6)
{
    float R[3][3], M[3][3];
    quat_to_rot(q, R);
    float sv[3] = {mod * s[0], mod * s[1], mod * s[2]};
    for (int i = 0; i < 3; i++) for (int a = 0; a < 3; a++) M[i][a] = sv[i] * R[a][i];
#define SIG(a, b) (M[0][a] * M[0][b] + M[1][a] * M[1][b] + M[2][a] * M[2][b])
    c6[0] = SIG(0, 0); c6[1] = SIG(0, 1); c6[2] = SIG(0, 2);
    c6[3] = SIG(1, 1); c6[4] = SIG(1, 2); c6[5] = SIG(2, 2);
#undef SIG
}

/* A = Jm * Rv (2x3), the affine approximation of the projection (forward.cu:112-143).
 * vm is the flat transposed view matrix: Rv[k][r] = vm[4r+k]. Also returns clamp masks. */
static void ewa_A(const float* mean, const float* vm, float fx, float fy, float tanx, float tany,
                  float A[2][3], float t_out[3], float* xmul, float* ymul)
{
    float t[3] = {vm[0] * mean[0] + vm[4] * mean[1] + vm[8] * mean[2] + vm[12],
                  vm[1] * mean[0] + vm[5] * mean[1] + vm[9] * mean[2] + vm[13],
                  vm[2] * mean[0] + vm[6] * mean[1] + vm[10] * mean[2] + vm[14]};
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf_(limx, fmaxf_(-limx, txtz)) * t[2];
    t[1] = fminf_(limy, fmaxf_(-limy, tytz)) * t[2];
    if (xmul) *xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    if (ymul) *ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float j00 = fx / t[2], j02 = -(fx * t[0]) / (t[2] * t[2]);
    const float j11 = fy / t[2], j12 = -(fy * t[1]) / (t[2] * t[2]);
    for (int r = 0; r < 3; r++) {
        const float rv0 = vm[4 * r + 0], rv1 = vm[4 * r + 1], rv2 = vm[4 * r + 2];
        A[0][r] = rv0 * j00 + rv1 * 0.0f + rv2 * j02;
        A[1][r] = rv0 * 0.0f + rv1 * j11 + rv2 * j12;
    }
    if (t_out) { t_out[0] = t[0]; t_out[1] = t[1]; t_out[2] = t[2]; }
}

static void sym3(const float* c6, float S[3][3])
{
    S[0][0] = c6[0]; S[0][1] = c6[1]; S[0][2] = c6[2];
    S[1][0] = c6[1]; S[1][1] = c6[3]; S[1][2] = c6[4];
    S[2][0] = c6[2]; S[2][1] = c6[4]; S[2][2] = c6[5];
}

/* cov2D = A Sigma A^T (+0.3 on the diagonal): forward.cu:144-150 */
static void cov2d_from(const float A[2][3], const float* c6, float* a, float* b, float* c)
{
    float S[3][3], SA[2][3];
    sym3(c6, S);
    for (int i = 0; i < 2; i++) for (int r = 0; r < 3; r++)
        SA[i][r] = S[r][0] * A[i][0] + S[r][1] * A[i][1] + S[r][2] * A[i][2];
    *a = A[0][0] * SA[0][0] + A[0][1] * SA[0][1] + A[0][2] * SA[0][2] + 0.3f;
    *b = A[0][0] * SA[1][0] + A[0][1] * SA[1][1] + A[0][2] * SA[1][2];
    *c = A[1][0] * SA[1][0] + A[1][1] * SA[1][1] + A[1][2] * SA[1][2] + 0.3f;
}

/* SH basis for degree <= 3 at unit direction d: forward.cu:68-97 (same as utils/sh_utils.py:74-100) */
static int sh_basis(int deg, const float* d, float* B)
{
    B[0] = kC0;
    if (deg < 1) return 1;
    const float x = d[0], y = d[1], z = d[2];
    B[1] = -kC1 * y; B[2] = kC1 * z; B[3] = -kC1 * x;
    if (deg < 2) return 4;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    B[4] = kC2[0] * xy; B[5] = kC2[1] * yz; B[6] = kC2[2] * (2.0f * zz - xx - yy);
    B[7] = kC2[3] * xz; B[8] = kC2[4] * (xx - yy);
    if (deg < 3) return 9;
    B[9] = kC3[0] * y * (3.0f * xx - yy);
    B[10] = kC3[1] * xy * z;
    B[11] = kC3[2] * y * (4.0f * zz - xx - yy);
    B[12] = kC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    B[13] = kC3[4] * x * (4.0f * zz - xx - yy);
    B[14] = kC3[5] * z * (xx - yy);
    B[15] = kC3[6] * x * (xx - 3.0f * yy);
    return 16;
}


/* ------------------------------------------------------------------------------------------
 * Tile culling (NOT in the reference -- an exactness-preserving optimisation of the HIP path,
 * restated here so that culled tile lists can be compared bit for bit; cull = 0 gives the
 * reference's AABB lists).  A Gaussian can only pass the alpha >= 1/255 test of the blend
 * (forward.cu:424-425) where  q = a dx^2 + 2 b dx dy + c dy^2 <= 2 ln(255 o).  Tiles of the
 * reference rectangle whose 16x16 pixel-centre box lies entirely outside that ellipse (with a
 * 0.1 % + 1e-3 safety margin, far above fp32 rounding of `power`) are dropped.  All arithmetic
 * is +,-,*,/,sqrt in fp32 so that gcc and hipcc (-ffp-contract=off) agree exactly.
 * ---------------------------------------------------------------------------------------- */
#if defined(__GNUC__) && !defined(__clang__)
#define ORC_DECISION __attribute__((noinline, optimize("fp-contract=off")))
#else
#define ORC_DECISION __attribute__((noinline))
#endif
static ORC_DECISION float ln_portable(float x_in)           /* |error| < 2e-6 for x >= 1; basic IEEE ops only */
{
    const orc_f32 x = (orc_f32)x_in;
    uint32_t u; memcpy(&u, &x, 4);
    const int e = (int)(u >> 23) - 127;
    u = (u & 0x007FFFFFu) | 0x3F800000u;
    orc_f32 m32; memcpy(&m32, &u, 4);
    const float m = m32;
    const float s = (m - 1.0f) / (m + 1.0f), z = s * s;
    const float poly = 1.0f + z * (0.33333334f + z * (0.2f + z * (0.14285715f + z * 0.11111111f)));
    return (float)e * 0.6931472f + 2.0f * s * poly;
}

static inline float clampf_(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Is the conic so close to singular that the reference's fp32 evaluation of `power` (forward.cu:419, backward.cu:644) can come out
 * POSITIVE through rounding?  |rounding| <= ~4 ulp of S = a dx^2 + c dy^2 while -power >= S (1 - |b| / sqrt(a c)) / 2, so that needs
 * b^2 > (1 - 1e-6) a c; the test below keeps a decade of margin.  Such Gaussians (needles hundreds of pixels long and half a pixel
 * wide) are exempt from the tile cull -- its margin is sized for well-conditioned conics -- and the HIP blend kernels evaluate them
 * with the reference's own expression, `power > 0` skip included (ibgs_amd/csrc/common.h: conic_is_risky). */
static inline int conic_is_risky(float A, float B, float C) { return B * B > 0.99999f * (A * C); }

/* statistics of the last orc_render_forward / orc_render_backward call: (pixel, Gaussian) pairs dropped by `power > 0` */
static long long g_power_skips[2] = {0, 0};
void orc_power_skips(long long* out) { out[0] = g_power_skips[0]; out[1] = g_power_skips[1]; }

/* Row by row: the part of the ellipse q <= qmax inside the band of pixel centres y in [Y0, Y1] of one tile row is convex, so the tiles it
 * reaches in that row are those whose centre box [16 tx, 16 tx + 15] meets its x-range -- one run of consecutive tiles.  For a fixed y
 * the ellipse spans x = (-B y -+ sqrt(A qmax - det y^2)) / A; the upper end is concave in y with its maximum at
 * y* = -B sqrt(qmax / (C det)) (the lower end at -y*), so over the band the extremes sit at y* clamped into the band.
 * ~30 operations per ROW instead of ~50 per tile; the same tiles as a closest-point test per tile, up to the 0.01 px slack below.
 * (ibgs_amd/csrc/common.h: CullRows / cull_rows_setup / cull_row_run -- the same operations in the same order.) */
typedef struct { float px, py, B, det, invA, aq, ymax, ystar; int x0, x1; } CullRows;
/* The tile cull takes DECISIONS, and two places take the same ones: the count in tile_cull and the emission in orc_bin.  They must agree in EVERY build of this
 * file -- in the fma-contracted twin gcc was free to contract A C - B B one way at one site and another way at the other, and its binning then emitted a
 * different number of entries than its preprocess had counted (orc_bin: -2).  So the helpers below are never contracted and never inlined into contracted code
 * (the HIP side does the same: `#pragma clang fp contract(off)` in common.h), and the determinant has one home. */
static ORC_DECISION float cull_det(float A, float B, float C) { return A * C - B * B; }
static ORC_DECISION void cull_rows_setup(CullRows* j, float px, float py, float A, float B, float C, float det, float qmax, int x0, int x1)
{
    j->px = px; j->py = py; j->B = B; j->det = det; j->x0 = x0; j->x1 = x1;
    j->invA = 1.0f / A; j->aq = A * qmax;
    j->ymax = sqrtf(j->aq / det); j->ystar = -B * sqrtf(qmax / (C * det));
}
/* tiles [*t0, *t1] of tile row ty (inside the tightened rectangle's columns [x0, x1)); returns 0 when the row holds none */
static ORC_DECISION int cull_row_run(const CullRows* j, int ty, int* t0_out, int* t1_out)
{
    const float Y0 = (float)(ty * 16) - j->py, Y1 = Y0 + 15.0f;
    const float yb0 = fmaxf_(Y0, -j->ymax), yb1 = fminf_(Y1, j->ymax);
    if (yb0 > yb1) return 0;
    const float yu = clampf_(j->ystar, yb0, yb1), yl = clampf_(-j->ystar, yb0, yb1);
    const float du = fmaxf_(j->aq - j->det * yu * yu, 0.0f), dl = fmaxf_(j->aq - j->det * yl * yl, 0.0f);
    const float xhi = (-j->B * yu + sqrtf(du)) * j->invA, xlo = (-j->B * yl - sqrtf(dl)) * j->invA;
    int t0 = (int)ceilf((xlo - 0.01f + j->px - 15.0f) / 16.0f), t1 = (int)floorf((xhi + 0.01f + j->px) / 16.0f);
    t0 = imax(t0, j->x0); t1 = imin(t1, j->x1 - 1);
    *t0_out = t0; *t1_out = t1;
    return t1 >= t0;
}
static ORC_DECISION float cull_qmax(float o) { return 2.0f * ln_portable(255.0f * o) * 1.001f + 0.001f; }

/* In: pixel centre, cov2D diagonal (with the 0.3), conic, opacity, reference rectangle.
 * Out: tightened rectangle and which of its tiles survive:
 *   <= CULL_MAX_TILES tiles: a bit mask (row-major inside the tightened rectangle, CULL_WORDS x 64 bits);
 *   more tiles: mask all ZERO = "the row runs decide" (orc_bin recomputes cull_row_run per tile row: rectangles of any size are culled),
 *               mask all ONES = every tile of the rectangle survives (near-singular or degenerate conics; cull = 0).
 * Returns the tile count. */
#define CULL_WORDS 4
#define CULL_MAX_TILES (64 * CULL_WORDS)
static uint32_t tile_cull(float px, float py, float sxx, float syy, float A, float B, float C, float o,
                          int* x0, int* y0, int* x1, int* y1, uint64_t* mask /* [CULL_WORDS] */)
{
    for (int k = 0; k < CULL_WORDS; k++) mask[k] = ~0ull;
    if (conic_is_risky(A, B, C)) return (uint32_t)((*x1 - *x0) * (*y1 - *y0));      /* keeps the reference rectangle, every tile of it */
    const float x255 = 255.0f * o;
    if (!(x255 >= 1.0f)) { *x1 = *x0; *y1 = *y0; for (int k = 0; k < CULL_WORDS; k++) mask[k] = 0; return 0; }
    const float qmax = cull_qmax(o);
    const float hx = sqrtf(qmax * sxx), hy = sqrtf(qmax * syy);
    int tx0 = (int)ceilf((px - hx - 15.0f) / 16.0f), tx1 = (int)floorf((px + hx) / 16.0f) + 1;
    int ty0 = (int)ceilf((py - hy - 15.0f) / 16.0f), ty1 = (int)floorf((py + hy) / 16.0f) + 1;
    tx0 = imax(tx0, *x0); tx1 = imin(tx1, *x1); ty0 = imax(ty0, *y0); ty1 = imin(ty1, *y1);
    if (tx1 <= tx0 || ty1 <= ty0) { *x1 = *x0; *y1 = *y0; for (int k = 0; k < CULL_WORDS; k++) mask[k] = 0; return 0; }
    *x0 = tx0; *x1 = tx1; *y0 = ty0; *y1 = ty1;
    const int w = tx1 - tx0, h = ty1 - ty0;
    const float det = cull_det(A, B, C);
    if (!(A > 0.0f) || !(C > 0.0f) || !(det > 0.0f)) return (uint32_t)(w * h);
    const int big = w * h > CULL_MAX_TILES;
    uint64_t m[CULL_WORDS] = {0, 0, 0, 0}; uint32_t cnt = 0;
    CullRows j;
    cull_rows_setup(&j, px, py, A, B, C, det, qmax, tx0, tx1);
    for (int ty = ty0; ty < ty1; ty++) {
        int t0, t1;
        if (!cull_row_run(&j, ty, &t0, &t1)) continue;
        cnt += (uint32_t)(t1 - t0 + 1);
        if (!big) for (int tx = t0; tx <= t1; tx++) { const int bit = (ty - ty0) * w + (tx - tx0); m[bit >> 6] |= 1ull << (bit & 63); }
    }
    for (int k = 0; k < CULL_WORDS; k++) mask[k] = m[k];          /* big: all zero = "rows decide" */
    return cnt;
}

/* ------------------------------------------------------------------------------------------
 * A1  preprocess (forward.cu:193-295, auxiliary.h:143-168). All outputs are zero for culled
 * Gaussians (the reference leaves them uninitialised).
 * Returns the number of Gaussians with radius > 0.
 * ---------------------------------------------------------------------------------------- */
int orc_preprocess(int P, int D, int M,
                   const float* means3D, const float* scales, float scale_modifier,
                   const float* rotations, const float* opacities, const float* shs,
                   const float* cov3D_precomp, const float* colors_precomp,
                   const float* vm, const float* pm, const float* campos,
                   int W, int H, float tanfovx, float tanfovy, int render_depth_only, int cull,
                   int32_t* radii, float* means2D, float* depths, float* cov3D, float* rgb,
                   float* conic_opacity, uint32_t* tiles_touched, uint8_t* clamped,
                   int32_t* rect4 /* P x 4: x0,y0,x1,y1 */, uint64_t* tmask)
{
    const float fy = H / (2.0f * tanfovy), fx = W / (2.0f * tanfovx); /* rasterizer_impl.cu:362-363 */
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    int visible = 0;
#pragma omp parallel for schedule(static) reduction(+ : visible)
    for (int i = 0; i < P; i++) {
        radii[i] = 0; tiles_touched[i] = 0; depths[i] = 0.f;
        rect4[4 * i] = rect4[4 * i + 1] = rect4[4 * i + 2] = rect4[4 * i + 3] = 0; for (int k = 0; k < CULL_WORDS; k++) tmask[CULL_WORDS * i + k] = 0;
        means2D[2 * i] = means2D[2 * i + 1] = 0.f;
        for (int k = 0; k < 6; k++) cov3D[6 * i + k] = 0.f;
        for (int k = 0; k < 3; k++) { rgb[3 * i + k] = 0.f; clamped[3 * i + k] = 0; }
        for (int k = 0; k < 4; k++) conic_opacity[4 * i + k] = 0.f;

        const float* p = means3D + 3 * i;
        const float hx = pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12];
        const float hy = pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13];
        const float hw = pm[3] * p[0] + pm[7] * p[1] + pm[11] * p[2] + pm[15];
        const float pw = 1.0f / (hw + 0.0000001f);
        const float zview = vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14];
        if (zview <= 0.2f) continue;                       /* near cull, auxiliary.h:158 */

        const float* c6;
        if (cov3D_precomp) c6 = cov3D_precomp + 6 * i;
        else { cov3d_from_scale_rot(scales + 3 * i, scale_modifier, rotations + 4 * i, cov3D + 6 * i); c6 = cov3D + 6 * i; }

        float A[2][3], a, b, c;
        ewa_A(p, vm, fx, fy, tanfovx, tanfovy, A, NULL, NULL, NULL);
        cov2d_from(A, c6, &a, &b, &c);
        const float det = a * c - b * b;
        if (det == 0.0f) continue;
        const float det_inv = 1.f / det;
        const float mid = 0.5f * (a + c);
        const float lam1 = mid + sqrtf(fmaxf_(0.1f, mid * mid - det));
        const float lam2 = mid - sqrtf(fmaxf_(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(fmaxf_(lam1, lam2)));
        const float px = ndc_to_pix(hx * pw, W), py = ndc_to_pix(hy * pw, H);
        int x0, y0, x1, y1;
        tile_rect(px, py, (int)my_radius, gx, gy, &x0, &y0, &x1, &y1);
        if ((x1 - x0) * (y1 - y0) == 0) continue;

        if (!colors_precomp && !render_depth_only) {       /* forward.cu:280-286, 58-109 */
            float d[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
            const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] /= len; d[1] /= len; d[2] /= len;
            float B[16];
            const int nb = sh_basis(D, d, B);
            const float* sh = shs + (size_t)i * M * 3;
            for (int ch = 0; ch < 3; ch++) {
                float r = B[0] * sh[ch];
                for (int k = 1; k < nb; k++) r = r + B[k] * sh[3 * k + ch];
                r += 0.5f;
                clamped[3 * i + ch] = (r < 0);
                rgb[3 * i + ch] = fmaxf_(r, 0.0f);
            }
        }
        depths[i] = zview;
        radii[i] = (int32_t)my_radius;
        means2D[2 * i] = px; means2D[2 * i + 1] = py;
        conic_opacity[4 * i + 0] = c * det_inv;
        conic_opacity[4 * i + 1] = -b * det_inv;
        conic_opacity[4 * i + 2] = a * det_inv;
        conic_opacity[4 * i + 3] = opacities[i];
        uint32_t nt = (uint32_t)((y1 - y0) * (x1 - x0));
        uint64_t mk[CULL_WORDS] = {~0ull, ~0ull, ~0ull, ~0ull};
        if (cull) nt = tile_cull(px, py, a, c, c * det_inv, -b * det_inv, a * det_inv, opacities[i], &x0, &y0, &x1, &y1, mk);
        tiles_touched[i] = nt; for (int k = 0; k < CULL_WORDS; k++) tmask[CULL_WORDS * i + k] = mk[k];
        rect4[4 * i] = x0; rect4[4 * i + 1] = y0; rect4[4 * i + 2] = x1; rect4[4 * i + 3] = y1;
        visible++;
    }
    return visible;
}

/* V1 checkFrustum: rasterizer_impl.cu:171-183 */
void orc_mark_visible(int P, const float* means3D, const float* vm, uint8_t* present)
{
    for (int i = 0; i < P; i++) {
        const float* p = means3D + 3 * i;
        present[i] = (vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14]) > 0.2f;
    }
}

/* ------------------------------------------------------------------------------------------
 * A2-A5 binning: tile/depth keys in emission order, stable LSD radix sort on the low
 * 32+bit bits, tile ranges (rasterizer_impl.cu:187-255, 426-466).
 * ---------------------------------------------------------------------------------------- */
int64_t orc_bin_count(int P, const uint32_t* tiles_touched)
{
    int64_t R = 0;
    for (int i = 0; i < P; i++) R += tiles_touched[i];
    return R;
}

int orc_bin(int P, int64_t R, const int32_t* radii, const int32_t* rect4, const uint64_t* tmask, const float* depths,
            const float* means2D, const float* conic_opacity /* rectangles above CULL_MAX_TILES tiles whose mask is zero: the row runs are recomputed */,
            int W, int H, uint64_t* keys_sorted, uint32_t* point_list, uint32_t* ranges /* tiles*2 */)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
    if (R == 0) return 0;
    uint64_t* k0 = (uint64_t*)malloc(sizeof(uint64_t) * R);
    uint32_t* v0 = (uint32_t*)malloc(sizeof(uint32_t) * R);
    uint32_t* v1 = point_list;
    uint64_t* k1 = keys_sorted;
    if (!k0 || !v0) { free(k0); free(v0); return -1; }
    int64_t off = 0;
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        const int x0 = rect4[4 * i], y0 = rect4[4 * i + 1], x1 = rect4[4 * i + 2], y1 = rect4[4 * i + 3];
        const int w = x1 - x0, dense = (w * (y1 - y0) > CULL_MAX_TILES);
        const orc_f32 depth32 = (orc_f32)depths[i];          /* the key carries the fp32 bits of the depth (rasterizer_impl.cu:216-220) */
        uint32_t dbits; memcpy(&dbits, &depth32, 4);
        if (dense && tmask[CULL_WORDS * i] == 0ull) {          /* culled row by row (tile_cull): the same runs, recomputed */
            const float A = conic_opacity[4 * i], B = conic_opacity[4 * i + 1], C = conic_opacity[4 * i + 2];
            CullRows j;
            cull_rows_setup(&j, means2D[2 * i], means2D[2 * i + 1], A, B, C, cull_det(A, B, C), cull_qmax(conic_opacity[4 * i + 3]), x0, x1);
            for (int y = y0; y < y1; y++) {
                int t0, t1;
                if (!cull_row_run(&j, y, &t0, &t1)) continue;
                for (int x = t0; x <= t1; x++) {
                    if (off < R) { k0[off] = ((uint64_t)(y * gx + x) << 32) | dbits; v0[off] = (uint32_t)i; }
                    off++;
                }
            }
            continue;
        }
        for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) {
            const int bit = (y - y0) * w + (x - x0);
            if (!dense && !((tmask[CULL_WORDS * i + (bit >> 6)] >> (bit & 63)) & 1ull)) continue;
            if (off < R) { k0[off] = ((uint64_t)(y * gx + x) << 32) | dbits; v0[off] = (uint32_t)i; }
            off++;
        }
    }
    if (off != R) { free(k0); free(v0); return -2; }
    const int nbits = 32 + (int)orc_higher_msb((uint32_t)(gx * gy));
    /* stable LSD radix, 8 bits per pass; ping-pong between (k0,v0) and (k1,v1) */
    uint64_t *ka = k0, *kb = k1; uint32_t *va = v0, *vb = v1;
    for (int shift = 0; shift < nbits; shift += 8) {
        int64_t cnt[257]; memset(cnt, 0, sizeof(cnt));
        const int width = (nbits - shift) < 8 ? (nbits - shift) : 8;
        const uint64_t mask = (1ull << width) - 1;
        for (int64_t j = 0; j < R; j++) cnt[((ka[j] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (int64_t j = 0; j < R; j++) {
            const int64_t dst = cnt[(ka[j] >> shift) & mask]++;
            kb[dst] = ka[j]; vb[dst] = va[j];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    if (ka != k1) { memcpy(k1, ka, sizeof(uint64_t) * R); memcpy(v1, va, sizeof(uint32_t) * R); }
    free(k0); free(v0);
    for (int64_t j = 0; j < R; j++) {            /* identifyTileRanges */
        const uint32_t cur = (uint32_t)(k1[j] >> 32);
        if (j == 0) ranges[2 * cur] = 0;
        else {
            const uint32_t prev = (uint32_t)(k1[j - 1] >> 32);
            if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)j; ranges[2 * cur] = (uint32_t)j; }
        }
        if (j == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Texture emulation (Appendix A.5 of SURVEY.md; rasterizer_impl.cu:117-130): layered,
 * unnormalised coordinates, clamp addressing, linear filter. quant != 0 rounds the two
 * filter weights to 8 fractional bits like the CUDA texture unit (1.8 fixed point).
 * ---------------------------------------------------------------------------------------- */
static inline float q8(float a, int quant) { return quant ? floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f) : a; }

static void tex_rgb(const float* img /* 3 planes HxW */, int W, int H, float x, float y, int quant, float out[3])
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fxi = floorf(xb), fyi = floorf(yb);
    const float a = q8(xb - fxi, quant), b = q8(yb - fyi, quant);
    const int i0 = imin(W - 1, imax(0, (int)fxi)), i1 = imin(W - 1, imax(0, (int)fxi + 1));
    const int j0 = imin(H - 1, imax(0, (int)fyi)), j1 = imin(H - 1, imax(0, (int)fyi + 1));
    const size_t hw = (size_t)W * H;
    for (int ch = 0; ch < 3; ch++) {
        const float* pl = img + ch * hw;
        const float t00 = pl[(size_t)j0 * W + i0], t10 = pl[(size_t)j0 * W + i1];
        const float t01 = pl[(size_t)j1 * W + i0], t11 = pl[(size_t)j1 * W + i1];
        out[ch] = (1.f - a) * (1.f - b) * t00 + a * (1.f - b) * t10 + (1.f - a) * b * t01 + a * b * t11;
    }
}

static float tex_1(const float* pl, int W, int H, float x, float y, int quant)
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fxi = floorf(xb), fyi = floorf(yb);
    const float a = q8(xb - fxi, quant), b = q8(yb - fyi, quant);
    const int i0 = imin(W - 1, imax(0, (int)fxi)), i1 = imin(W - 1, imax(0, (int)fxi + 1));
    const int j0 = imin(H - 1, imax(0, (int)fyi)), j1 = imin(H - 1, imax(0, (int)fyi + 1));
    const float t00 = pl[(size_t)j0 * W + i0], t10 = pl[(size_t)j0 * W + i1];
    const float t01 = pl[(size_t)j1 * W + i0], t11 = pl[(size_t)j1 * W + i1];
    return (1.f - a) * (1.f - b) * t00 + a * (1.f - b) * t10 + (1.f - a) * b * t01 + a * b * t11;
}

/* ------------------------------------------------------------------------------------------
 * F1-F3 render forward, one pixel at a time (forward.cu:303-665).
 * Per-pixel state arrays are sized HW; slot-k planes are k*HW + pix.
 * ---------------------------------------------------------------------------------------- */
void orc_render_forward(
    int W, int H, const uint32_t* ranges, const uint32_t* point_list,
    const float* means2D, const float* features, const float* all_map, const float* conic_opacity,
    const float* vm, const float* campos, const float* bg, float tanfovx, float tanfovy,
    int n_src, const float* ref_to_src, const float* src_cam_pos, const float* src_images,
    const float* src_depths, int L, float depth_thr, int render_geo, int depth_only, int tex_quant,
    float* final_T, uint32_t* n_contrib, float* cache_sum_w, uint32_t* cache_low, uint32_t* cache_high,
    int32_t* valid_src_idx, float* valid_src_w,
    float* out_color, float* out_normal, float* out_depth, float* out_cam_feat, float* out_warped,
    float* out_min_depth_diff, float* out_camera_ray, int32_t* out_mask)
{
    const float fy = H / (2.0f * tanfovy), fx = W / (2.0f * tanfovx);
    const float cx = (float)(W * 0.5f), cy = (float)(H * 0.5f);  /* rasterizer_impl.cu:477 */
    const int gx = (W + TILE - 1) / TILE;
    const size_t HW = (size_t)W * H;
    const float eps = 1.0e-8f;
    const float inv_fx = 1.0f / fx, inv_fy = 1.0f / fy;
    const int before_cap = (L % 2 == 0) ? (L / 2) : ((L + 1) / 2);
    const int below_cap = L - before_cap;

    long long nskip = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : nskip)
    for (int py = 0; py < H; py++) for (int px = 0; px < W; px++) {
        const size_t pix = (size_t)py * W + px;
        const int tile = (py / TILE) * gx + (px / TILE);
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int n = (int)(r1 - r0);
        const float pixx = (float)px, pixy = (float)py;
        const float rayx = (pixx - cx) / fx, rayy = (pixy - cy) / fy;
        const float pdx = pixx - cx, pdy = pixy - cy;

        float T = 1.0f, C[3] = {0, 0, 0}, N[3] = {0, 0, 0};
        uint32_t contributor = 0, last_contributor = 0;
        float buf_d[MAX_L] = {0}, buf_w[MAX_L] = {0}; uint32_t buf_c[MAX_L] = {0};
        int before_ptr = 0, below_count = 0;
        float tot_w = 0.f, wd_sum = 0.f;
        int done = 0;

        for (int k = 0; k < n && !done; k++) {
            contributor++;
            const uint32_t id = point_list[r0 + k];
            const float dx = means2D[2 * id] - pixx, dy = means2D[2 * id + 1] - pixy;
            const float* co = conic_opacity + 4 * id;
            const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
            if (power > 0.0f) { nskip++; continue; }
            const float alpha = fminf_(0.99f, co[3] * expf(power));   /* reference: __expf (Q1) */
            if (alpha < 1.0f / 255.0f) continue;
            const float test_T = T * (1.0f - alpha);
            if (test_T < 0.0001f) { done = 1; continue; }
            const float aT = alpha * T;
            if (!depth_only) for (int ch = 0; ch < 3; ch++) C[ch] += features[3 * id + ch] * aT;

            float dep = 0.0f;
            if (render_geo || depth_only) {
                const float* am = all_map + 5 * id;
                dep = -am[4] / (am[0] * rayx + am[1] * rayy + am[2] + eps);
            }
            if (render_geo) {
                const float* am = all_map + 5 * id;
                for (int ch = 0; ch < 3; ch++) N[ch] += am[ch] * aT;
                if (dep > 0.0f) {
                    if (T > 0.5f) {
                        buf_d[before_ptr] = dep; buf_w[before_ptr] = aT; buf_c[before_ptr] = contributor;
                        before_ptr = (before_ptr + 1) % before_cap;
                    } else if (below_count < below_cap) {
                        const int s = before_cap + below_count;
                        buf_d[s] = dep; buf_w[s] = aT; buf_c[s] = contributor;
                        below_count++;
                    }
                }
            }
            if (depth_only && dep > 0.0f) {            /* forward.cu:466-489 */
                if (T > 0.5f) {
                    const int s = before_ptr;
                    tot_w -= buf_w[s]; wd_sum -= buf_w[s] * buf_d[s];
                    buf_d[s] = dep; buf_w[s] = aT;
                    before_ptr = (before_ptr + 1) % before_cap;
                    tot_w += aT; wd_sum += aT * dep;
                } else if (below_count < below_cap) {
                    const int s = before_cap + below_count;
                    buf_d[s] = dep; buf_w[s] = aT; below_count++;
                    tot_w += aT; wd_sum += aT * dep;
                }
                if (below_count == below_cap) {
                    /* 'break' leaves only the current 256-entry round; the pixel resumes at the
                     * next round start without counting the entries it skipped. */
                    T = test_T; last_contributor = contributor;
                    k = (k / ROUND + 1) * ROUND - 1;
                    continue;
                }
            }
            T = test_T; last_contributor = contributor;
        }

        final_T[pix] = T; n_contrib[pix] = last_contributor;
        if (!depth_only) for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix] = C[ch] + T * bg[ch];
        if (depth_only) out_depth[pix] = wd_sum / (tot_w + eps);

        if (render_geo) {                                   /* forward.cu:512-663 */
            float tw = 0.f, tw_src[MAX_SRC] = {0}, wc[MAX_SRC * 3] = {0}, med = 0.f;
            uint32_t lo = buf_c[0], hi = buf_c[0];          /* Q4: slot 0 even if empty */
            for (int s = 0; s < L; s++) {
                const float w = buf_w[s];
                if (w == 0.0f) continue;
                const float d = buf_d[s];
                const float X = pdx * d * inv_fx, Y = pdy * d * inv_fy, Z = d;
                for (int si = 0; si < n_src; si++) {
                    const float* r = ref_to_src + 16 * si;
                    const float tx = r[0] * X + r[1] * Y + r[2] * Z + r[3] * 1.0f;
                    const float ty = r[4] * X + r[5] * Y + r[6] * Z + r[7] * 1.0f;
                    const float tz = r[8] * X + r[9] * Y + r[10] * Z + r[11] * 1.0f;
                    const float iz = 1.0f / (tz + eps);
                    const float u = tx * fx * iz + cx, v = ty * fy * iz + cy;
                    if (u >= 0.0f && u <= (float)(W - 1) && v >= 0.0f && v <= (float)(H - 1)) {
                        float col[3];
                        tex_rgb(src_images + (size_t)si * 3 * HW, W, H, u + 0.5f, v + 0.5f, tex_quant, col);
                        for (int ch = 0; ch < 3; ch++) wc[3 * si + ch] += w * col[ch];
                        tw_src[si] += w;
                    }
                }
                tw += w; med += w * d;
                if (buf_c[s] < lo) lo = buf_c[s];
                if (buf_c[s] > hi) hi = buf_c[s];
            }
            cache_low[pix] = lo; cache_high[pix] = hi; cache_sum_w[pix] = tw;
            med /= (tw + eps);
            const float mX = pdx * med * inv_fx, mY = pdy * med * inv_fy, mZ = med;
            const float qx = mX - vm[12], qy = mY - vm[13], qz = mZ - vm[14];
            const float wx = vm[0] * qx + vm[1] * qy + vm[2] * qz;
            const float wy = vm[4] * qx + vm[5] * qy + vm[6] * qz;
            const float wz = vm[8] * qx + vm[9] * qy + vm[10] * qz;
            float rd[3] = {wx - campos[0], wy - campos[1], wz - campos[2]};
            const float rl = sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]) + eps;
            rd[0] /= rl; rd[1] /= rl; rd[2] /= rl;
            for (int ch = 0; ch < 3; ch++) out_camera_ray[ch * HW + pix] = rd[ch];

            int nvalid = 0; float min_err = 1.0f;
            for (int si = 0; si < n_src; si++) {
                const float* r = ref_to_src + 16 * si;
                const float tx = r[0] * mX + r[1] * mY + r[2] * mZ + r[3] * 1.0f;
                const float ty = r[4] * mX + r[5] * mY + r[6] * mZ + r[7] * 1.0f;
                const float tz = r[8] * mX + r[9] * mY + r[10] * mZ + r[11] * 1.0f;
                const float iz = 1.0f / (tz + eps);
                const float u = tx * fx * iz + cx, v = ty * fy * iz + cy;
                float wdep = 0.0f;
                if (u >= 0.0f && u <= (float)(W - 1) && v >= 0.0f && v <= (float)(H - 1))
                    wdep = tex_1(src_depths + (size_t)si * HW, W, H, u + 0.5f, v + 0.5f, tex_quant);
                const float err = fabsf(wdep - tz) * iz;
                if (wdep > 0.0f && err < depth_thr) {
                    const float iw = 1.0f / (tw_src[si] + eps);
                    for (int ch = 0; ch < 3; ch++) {
                        wc[3 * si + ch] *= iw;
                        out_cam_feat[((size_t)nvalid * 4 + ch) * HW + pix] = campos[ch] - src_cam_pos[3 * si + ch];
                        out_warped[((size_t)nvalid * 3 + ch) * HW + pix] = wc[3 * si + ch];
                    }
                    float sd[3] = {wx - src_cam_pos[3 * si], wy - src_cam_pos[3 * si + 1], wz - src_cam_pos[3 * si + 2]};
                    const float sl = sqrtf(sd[0] * sd[0] + sd[1] * sd[1] + sd[2] * sd[2]) + eps;
                    sd[0] /= sl; sd[1] /= sl; sd[2] /= sl;
                    out_cam_feat[((size_t)nvalid * 4 + 3) * HW + pix] = sd[0] * rd[0] + sd[1] * rd[1] + sd[2] * rd[2];
                    if (si == 0) out_mask[pix] = 1;
                    valid_src_idx[(size_t)nvalid * HW + pix] = si;
                    valid_src_w[(size_t)nvalid * HW + pix] = tw_src[si];
                    nvalid++;
                    min_err = fminf_(min_err, err);
                    if (nvalid == MAX_SRC) break;
                }
            }
            if (nvalid <= MAX_SRC - 1) valid_src_idx[(size_t)nvalid * HW + pix] = -1;
            out_min_depth_diff[pix] = min_err;
            out_depth[pix] = med;
            for (int ch = 0; ch < 3; ch++) out_normal[ch * HW + pix] = N[ch];
        }
    }
    g_power_skips[0] = nskip;
}

/* Q3: derivative of the warped colour w.r.t. (u,v) as the reference computes it
 * (backward.cu:55-109): 4 linear-filtered fetches at integer coordinates. */
static void warp_grad_uv(const float* img, int W, int H, float u_in, float v_in, const float g[3],
                         int quant, float* du, float* dv)
{
    const float u = u_in + 0.5f, v = v_in + 0.5f;
    const int u0 = (int)floorf(u), v0 = (int)floorf(v), u1 = u0 + 1, v1 = v0 + 1;
    const float fu = u - (float)u0, fv = v - (float)v0, fu1 = 1.0f - fu, fv1 = 1.0f - fv;
    float I00[3], I01[3], I10[3], I11[3];
    tex_rgb(img, W, H, (float)u0, (float)v0, quant, I00);
    tex_rgb(img, W, H, (float)u1, (float)v0, quant, I01);
    tex_rgb(img, W, H, (float)u0, (float)v1, quant, I10);
    tex_rgb(img, W, H, (float)u1, (float)v1, quant, I11);
    float su = 0.f, sv = 0.f;
    float dIu[3], dIv[3];
    for (int ch = 0; ch < 3; ch++) {
        dIu[ch] = -fv1 * I00[ch] + fv1 * I01[ch] - fv * I10[ch] + fv * I11[ch];
        dIv[ch] = -fu1 * I00[ch] - fu * I01[ch] + fu1 * I10[ch] + fu * I11[ch];
    }
    su = g[0] * dIu[0] + g[1] * dIu[1] + g[2] * dIu[2];
    sv = g[0] * dIv[0] + g[1] * dIv[1] + g[2] * dIv[2];
    *du = su; *dv = sv;
}

/* ------------------------------------------------------------------------------------------
 * B1-B2 render backward (backward.cu:496-807). Accumulators are double, P-sized, zeroed
 * by the caller: acc_mean2D[P*2], acc_mean2D_abs[P*2], acc_conic[P*3] (x,y,w of the
 * float4), acc_opacity[P], acc_color[P*3], acc_all_map[P*5].
 * ---------------------------------------------------------------------------------------- */
void orc_render_backward(
    int W, int H, const uint32_t* ranges, const uint32_t* point_list,
    const float* means2D, const float* conic_opacity, const float* colors, const float* all_map,
    const float* bg, float tanfovx, float tanfovy,
    int n_src, const float* ref_to_src, const float* src_images,
    const float* depth_pixels, const float* warped_pixels,
    const float* final_T, const uint32_t* n_contrib, const float* cache_sum_w,
    const uint32_t* cache_low, const uint32_t* cache_high,
    const int32_t* valid_src_idx, const float* valid_src_w,
    const float* dL_dpix, const float* dL_dnormal, const float* dL_ddepth, const float* dL_dwarped,
    int render_geo, int tex_quant,
    double* acc_mean2D, double* acc_mean2D_abs, double* acc_conic, double* acc_opacity,
    double* acc_color, double* acc_all_map)
{
    (void)n_src;
    const float fy = H / (2.0f * tanfovy), fx = W / (2.0f * tanfovx);
    const int gx = (W + TILE - 1) / TILE;
    const size_t HW = (size_t)W * H;
    const float cx = (float)(W * 0.5f), cy = (float)(H * 0.5f);
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);

    /* rows in parallel; the double accumulators are updated atomically (sum order is then
     * nondeterministic at the 1e-16 level, far below the fp32 results being checked) */
    long long nskip = 0;
#ifndef ORC_ACC_FLOAT
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : nskip)
#endif
    for (int py = 0; py < H; py++) for (int px = 0; px < W; px++) {
        const size_t pix = (size_t)py * W + px;
        const int tile = (py / TILE) * gx + (px / TILE);
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int n = (int)(r1 - r0);
        const float pixx = (float)px, pixy = (float)py;
        /* dbl: (pixf.x - W * 0.5) / fx, backward.cu:545 */
        const float rayx = (float)(((double)pixx - W * 0.5) / (double)fx);
        const float rayy = (float)(((double)pixy - H * 0.5) / (double)fy);

        const float T_final = final_T[pix];
        float T = T_final;
        const int last_contributor = (int)n_contrib[pix];
        const int min_med = render_geo ? (int)cache_low[pix] : 0;
        const int max_med = render_geo ? (int)cache_high[pix] : 0;
        float accum_rec[3] = {0, 0, 0}, accum_n[3] = {0, 0, 0};
        float last_alpha = 0.f, last_color[3] = {0, 0, 0}, last_n[3] = {0, 0, 0};
        float g_pix[3], g_n[3] = {0, 0, 0}, g_d = 0.f;
        for (int ch = 0; ch < 3; ch++) g_pix[ch] = dL_dpix[ch * HW + pix];
        if (render_geo) {
            for (int ch = 0; ch < 3; ch++) g_n[ch] = dL_dnormal[ch * HW + pix];
            g_d = dL_ddepth[pix];
        }
        float bg_dot = 0.f;
        for (int ch = 0; ch < 3; ch++) bg_dot += bg[ch] * g_pix[ch];

        for (int k = n - 1; k >= 0; k--) {
            const uint32_t contributor = (uint32_t)k;       /* 0-based position in the list */
            if (contributor >= (uint32_t)last_contributor) continue;
            const uint32_t id = point_list[r0 + k];
            const float dx = means2D[2 * id] - pixx, dy = means2D[2 * id + 1] - pixy;
            const float* co = conic_opacity + 4 * id;
            const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
            if (power > 0.0f) { nskip++; continue; }
            const float G = expf(power);
            const float alpha = fminf_(0.99f, co[3] * G);
            if (alpha < 1.0f / 255.0f) continue;
            T = T / (1.f - alpha);
            const float w = alpha * T;                      /* dchannel_dcolor */

            float dL_dalpha = 0.0f;
            for (int ch = 0; ch < 3; ch++) {
                const float c = colors[3 * id + ch];
                accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                last_color[ch] = c;
                dL_dalpha += (c - accum_rec[ch]) * g_pix[ch];
                ACC_ADD(acc_color[3 * id + ch], (w * g_pix[ch]));
            }
            if (render_geo) {
                const float* am = all_map + 5 * id;
                float gm[5] = {0, 0, 0, 0, 0};
                for (int ch = 0; ch < 3; ch++) {
                    const float c = am[ch];
                    accum_n[ch] = last_alpha * last_n[ch] + (1.f - last_alpha) * accum_n[ch];
                    last_n[ch] = c;
                    dL_dalpha += (c - accum_n[ch]) * g_n[ch];
                    gm[ch] += w * g_n[ch];
                }
                /* unsigned comparison: min_med == 0 disables the branch (Q4), backward.cu:693 */
                if ((contributor >= (uint32_t)(min_med - 1)) && (contributor <= (uint32_t)(max_med - 1))) {
                    const float nx = am[0], ny = am[1], nz = am[2], dist = am[4];
                    const float tmp = (float)((double)(nx * rayx + ny * rayy + nz) + 1.0e-8);      /* dbl */
                    const float tmp2 = dist / (tmp * tmp);
                    const float dep = (float)(-(double)dist / ((double)(nx * rayx + ny * rayy + nz) + 1.0e-8)); /* dbl */
                    if (dep > 0.0f) {
                        const float X = (pixx - cx) * dep / fx, Y = (pixy - cy) * dep / fy, Z = dep;
                        const float sumw = cache_sum_w[pix];
                        float gdep = g_d * w / sumw;
                        dL_dalpha += g_d * (dep - depth_pixels[pix]) / sumw;
                        for (int m = 0; m < MAX_SRC; m++) {
                            const int si = valid_src_idx[(size_t)m * HW + pix];
                            if (si == -1) break;
                            const float* r = ref_to_src + 16 * si;
                            const float tx = r[0] * X + r[1] * Y + r[2] * Z + r[3];
                            const float ty = r[4] * X + r[5] * Y + r[6] * Z + r[7];
                            const float tz = r[8] * X + r[9] * Y + r[10] * Z + r[11];
                            const float u = (tx * fx / tz) + cx, v = (ty * fy / tz) + cy;
                            if (u >= 0 && u <= W - 1 && v >= 0 && v <= H - 1) {
                                float col[3], gc[3];
                                const float* img = src_images + (size_t)si * 3 * HW;
                                tex_rgb(img, W, H, u + 0.5f, v + 0.5f, tex_quant, col);
                                const float sw = valid_src_w[(size_t)m * HW + pix];
                                for (int ch = 0; ch < 3; ch++) {
                                    const float gw = dL_dwarped[((size_t)m * 3 + ch) * HW + pix];
                                    gc[ch] = gw * w / sw;
                                    dL_dalpha += gw * (col[ch] - warped_pixels[((size_t)m * 3 + ch) * HW + pix]) / sw;
                                }
                                const float Av = (pixx - cx) / fx, Bv = (pixy - cy) / fy;
                                const float U = r[0] * Av + r[1] * Bv + r[2];
                                const float V = r[4] * Av + r[5] * Bv + r[6];
                                const float Wc = r[8] * Av + r[9] * Bv + r[10];
                                const float den = (Wc * dep + r[11]);
                                const float dpx = fx * (U * r[11] - Wc * r[3]) / (den * den);
                                const float dpy = fy * (V * r[11] - Wc * r[7]) / (den * den);
                                float du, dv;
                                warp_grad_uv(img, W, H, u, v, gc, tex_quant, &du, &dv);
                                gdep += du * dpx + dv * dpy;
                                /* Q2: plane gradient is emitted inside the per-source in-bounds branch */
                                gm[4] += (-gdep / tmp);
                                gm[0] += gdep * tmp2 * rayx;
                                gm[1] += gdep * tmp2 * rayy;
                                gm[2] += gdep * tmp2;
                            }
                        }
                    }
                }
                for (int ch = 0; ch < 5; ch++) {
                    ACC_ADD(acc_all_map[5 * id + ch], gm[ch]);
                }
            }
            dL_dalpha *= T;
            last_alpha = alpha;
            dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;

            const float dL_dG = co[3] * dL_dalpha;
            const float gdx = G * dx, gdy = G * dy;
            const float dG_ddelx = -gdx * co[0] - gdy * co[1];
            const float dG_ddely = -gdy * co[2] - gdx * co[1];
            const float mx = dL_dG * dG_ddelx * ddelx_dx, my = dL_dG * dG_ddely * ddely_dy;
            ACC_ADD(acc_mean2D[2 * id], mx);
            ACC_ADD(acc_mean2D[2 * id + 1], my);
            ACC_ADD(acc_mean2D_abs[2 * id], fabsf(mx));
            ACC_ADD(acc_mean2D_abs[2 * id + 1], fabsf(my));
#ifdef ORC_LFORM
            /* diagnostic (oracle.variant("lform")): NOT the reference's arithmetic -- the conic sums in the well-conditioned form the HIP path uses for
             * near-singular conics (csrc/render_bwd.hip): dL/dcov2D = 0.5 sum q l l^T with l = conic d, which the chain below then takes as it is */
            {
                const float lx = co[0] * dx + co[1] * dy, ly = co[1] * dx + co[2] * dy, q = G * dL_dG;
                ACC_ADD(acc_conic[3 * id + 0], (0.5f * q * lx * lx));
                ACC_ADD(acc_conic[3 * id + 1], (q * lx * ly));
                ACC_ADD(acc_conic[3 * id + 2], (0.5f * q * ly * ly));
            }
#else
            ACC_ADD(acc_conic[3 * id + 0], (-0.5f * gdx * dx * dL_dG));
            ACC_ADD(acc_conic[3 * id + 1], (-0.5f * gdx * dy * dL_dG));
            ACC_ADD(acc_conic[3 * id + 2], (-0.5f * gdy * dy * dL_dG));
#endif
            ACC_ADD(acc_opacity[id], (G * dL_dalpha));
        }
    }
    g_power_skips[1] = nskip;
}

/* ------------------------------------------------------------------------------------------
 * B3 + B4 per-Gaussian backward (backward.cu:241-371, 443-493, 116-235, 375-438).
 * Inputs dL_dmean2D[P*3] (x,y used), dL_dconic[P*4] (x,y,w used), dL_dcolor[P*3] as float.
 * Outputs are fully written (zeros for radii <= 0).
 * ---------------------------------------------------------------------------------------- */
void orc_preprocess_backward(
    int P, int D, int M, const float* means3D, const int32_t* radii, const float* shs,
    const uint8_t* clamped, const float* scales, const float* rotations, float scale_modifier,
    const float* cov3D /* precomp or computed */, const float* vm, const float* pm, const float* campos,
    int W, int H, float tanfovx, float tanfovy,
    const float* dL_dmean2D, const float* dL_dconic, const float* dL_dcolor,
    float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot)
{
    const float fy = H / (2.0f * tanfovy), fx = W / (2.0f * tanfovx);
    memset(dL_dmean3D, 0, sizeof(float) * 3 * (size_t)P);
    memset(dL_dcov3D, 0, sizeof(float) * 6 * (size_t)P);
    if (shs) memset(dL_dsh, 0, sizeof(float) * 3 * (size_t)M * P);
    if (scales) { memset(dL_dscale, 0, sizeof(float) * 3 * (size_t)P); memset(dL_drot, 0, sizeof(float) * 4 * (size_t)P); }

#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (!(radii[i] > 0)) continue;
        const float* mean = means3D + 3 * i;
        const float* c6 = cov3D + 6 * i;
        /* ---- cov2D backward ---- */
        float A[2][3], t[3], xmul, ymul, a, b, c;
        ewa_A(mean, vm, fx, fy, tanfovx, tanfovy, A, t, &xmul, &ymul);
        cov2d_from(A, c6, &a, &b, &c);
        const float gcx = dL_dconic[4 * i], gcy = dL_dconic[4 * i + 1], gcz = dL_dconic[4 * i + 3];
        const float denom = a * c - b * b;
        float da = 0, db = 0, dc = 0;
        const float d2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* gS = dL_dcov3D + 6 * i;
        if (d2inv != 0) {
#ifdef ORC_LFORM
            { const float kk = d2inv * (denom * denom); da = kk * gcx; db = kk * gcy; dc = kk * gcz; }
#else
            da = d2inv * (-c * c * gcx + 2 * b * c * gcy + (denom - a * c) * gcz);
            dc = d2inv * (-a * a * gcz + 2 * a * b * gcy + (denom - a * c) * gcx);
            db = d2inv * 2 * (b * c * gcx - (denom + 2 * b * b) * gcy + a * b * gcz);
#endif
            gS[0] = (A[0][0] * A[0][0] * da + A[0][0] * A[1][0] * db + A[1][0] * A[1][0] * dc);
            gS[3] = (A[0][1] * A[0][1] * da + A[0][1] * A[1][1] * db + A[1][1] * A[1][1] * dc);
            gS[5] = (A[0][2] * A[0][2] * da + A[0][2] * A[1][2] * db + A[1][2] * A[1][2] * dc);
            gS[1] = 2 * A[0][0] * A[0][1] * da + (A[0][0] * A[1][1] + A[0][1] * A[1][0]) * db + 2 * A[1][0] * A[1][1] * dc;
            gS[2] = 2 * A[0][0] * A[0][2] * da + (A[0][0] * A[1][2] + A[0][2] * A[1][0]) * db + 2 * A[1][0] * A[1][2] * dc;
            gS[4] = 2 * A[0][2] * A[0][1] * da + (A[0][1] * A[1][2] + A[0][2] * A[1][1]) * db + 2 * A[1][1] * A[1][2] * dc;
        }
        float S[3][3]; sym3(c6, S);
        float dA[2][3];
        for (int r = 0; r < 3; r++) {
            const float a0S = A[0][0] * S[r][0] + A[0][1] * S[r][1] + A[0][2] * S[r][2];
            const float a1S = A[1][0] * S[r][0] + A[1][1] * S[r][1] + A[1][2] * S[r][2];
            dA[0][r] = 2 * a0S * da + a1S * db;
            dA[1][r] = 2 * a1S * dc + a0S * db;
        }
        /* Rv[k][r] = vm[4r+k]; dJ00 = sum_r Rv[0][r] dA0r etc. */
        const float dJ00 = vm[0] * dA[0][0] + vm[4] * dA[0][1] + vm[8] * dA[0][2];
        const float dJ02 = vm[2] * dA[0][0] + vm[6] * dA[0][1] + vm[10] * dA[0][2];
        const float dJ11 = vm[1] * dA[1][0] + vm[5] * dA[1][1] + vm[9] * dA[1][2];
        const float dJ12 = vm[2] * dA[1][0] + vm[6] * dA[1][1] + vm[10] * dA[1][2];
        const float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = xmul * -fx * tz2 * dJ02;
        const float dty = ymul * -fy * tz2 * dJ12;
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * t[0]) * tz3 * dJ02 + (2 * fy * t[1]) * tz3 * dJ12;
        float gm[3] = {vm[0] * dtx + vm[1] * dty + vm[2] * dtz,
                       vm[4] * dtx + vm[5] * dty + vm[6] * dtz,
                       vm[8] * dtx + vm[9] * dty + vm[10] * dtz};   /* assigned, backward.cu:370 */

        /* ---- projection of the 2D mean gradient (backward.cu:467-484) ---- */
        const float hw = pm[3] * mean[0] + pm[7] * mean[1] + pm[11] * mean[2] + pm[15];
        const float mw = 1.0f / (hw + 0.0000001f);
        const float mul1 = (pm[0] * mean[0] + pm[4] * mean[1] + pm[8] * mean[2] + pm[12]) * mw * mw;
        const float mul2 = (pm[1] * mean[0] + pm[5] * mean[1] + pm[9] * mean[2] + pm[13]) * mw * mw;
        const float g2x = dL_dmean2D[3 * i], g2y = dL_dmean2D[3 * i + 1];
        gm[0] += (pm[0] * mw - pm[3] * mul1) * g2x + (pm[1] * mw - pm[3] * mul2) * g2y;
        gm[1] += (pm[4] * mw - pm[7] * mul1) * g2x + (pm[5] * mw - pm[7] * mul2) * g2y;
        gm[2] += (pm[8] * mw - pm[11] * mul1) * g2x + (pm[9] * mw - pm[11] * mul2) * g2y;

        /* ---- SH backward (backward.cu:116-235) ---- */
        if (shs) {
            const float dorig[3] = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
            const float len = sqrtf(dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2]);
            const float d[3] = {dorig[0] / len, dorig[1] / len, dorig[2] / len};
            const float* sh = shs + (size_t)i * M * 3;
            float* gsh = dL_dsh + (size_t)i * M * 3;
            float g[3];
            for (int ch = 0; ch < 3; ch++) g[ch] = dL_dcolor[3 * i + ch] * (clamped[3 * i + ch] ? 0.f : 1.f);
            float B[16];
            const int nb = sh_basis(D, d, B);
            for (int k = 0; k < nb; k++) for (int ch = 0; ch < 3; ch++) gsh[3 * k + ch] = B[k] * g[ch];
            float dx3[3] = {0, 0, 0}, dy3[3] = {0, 0, 0}, dz3[3] = {0, 0, 0};
            const float x = d[0], y = d[1], z = d[2];
#define SHK(k, ch) sh[3 * (k) + (ch)]
            if (D > 0) for (int ch = 0; ch < 3; ch++) {
                dx3[ch] = -kC1 * SHK(3, ch); dy3[ch] = -kC1 * SHK(1, ch); dz3[ch] = kC1 * SHK(2, ch);
                if (D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    dx3[ch] += kC2[0] * y * SHK(4, ch) + kC2[2] * 2.f * -x * SHK(6, ch) + kC2[3] * z * SHK(7, ch) + kC2[4] * 2.f * x * SHK(8, ch);
                    dy3[ch] += kC2[0] * x * SHK(4, ch) + kC2[1] * z * SHK(5, ch) + kC2[2] * 2.f * -y * SHK(6, ch) + kC2[4] * 2.f * -y * SHK(8, ch);
                    dz3[ch] += kC2[1] * y * SHK(5, ch) + kC2[2] * 2.f * 2.f * z * SHK(6, ch) + kC2[3] * x * SHK(7, ch);
                    if (D > 2) {
                        dx3[ch] += (kC3[0] * SHK(9, ch) * 3.f * 2.f * xy + kC3[1] * SHK(10, ch) * yz + kC3[2] * SHK(11, ch) * -2.f * xy +
                                    kC3[3] * SHK(12, ch) * -3.f * 2.f * xz + kC3[4] * SHK(13, ch) * (-3.f * xx + 4.f * zz - yy) +
                                    kC3[5] * SHK(14, ch) * 2.f * xz + kC3[6] * SHK(15, ch) * 3.f * (xx - yy));
                        dy3[ch] += (kC3[0] * SHK(9, ch) * 3.f * (xx - yy) + kC3[1] * SHK(10, ch) * xz + kC3[2] * SHK(11, ch) * (-3.f * yy + 4.f * zz - xx) +
                                    kC3[3] * SHK(12, ch) * -3.f * 2.f * yz + kC3[4] * SHK(13, ch) * -2.f * xy +
                                    kC3[5] * SHK(14, ch) * -2.f * yz + kC3[6] * SHK(15, ch) * -3.f * 2.f * xy);
                        dz3[ch] += (kC3[1] * SHK(10, ch) * xy + kC3[2] * SHK(11, ch) * 4.f * 2.f * yz + kC3[3] * SHK(12, ch) * 3.f * (2.f * zz - xx - yy) +
                                    kC3[4] * SHK(13, ch) * 4.f * 2.f * xz + kC3[5] * SHK(14, ch) * (xx - yy));
                    }
                }
            }
#undef SHK
            const float gd[3] = {dx3[0] * g[0] + dx3[1] * g[1] + dx3[2] * g[2],
                                 dy3[0] * g[0] + dy3[1] * g[1] + dy3[2] * g[2],
                                 dz3[0] * g[0] + dz3[1] * g[1] + dz3[2] * g[2]};
            /* dnormvdv, auxiliary.h:111-121 */
            const float s2 = dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2];
            const float inv32 = 1.0f / sqrtf(s2 * s2 * s2);
            gm[0] += ((+s2 - dorig[0] * dorig[0]) * gd[0] - dorig[1] * dorig[0] * gd[1] - dorig[2] * dorig[0] * gd[2]) * inv32;
            gm[1] += (-dorig[0] * dorig[1] * gd[0] + (s2 - dorig[1] * dorig[1]) * gd[1] - dorig[2] * dorig[1] * gd[2]) * inv32;
            gm[2] += (-dorig[0] * dorig[2] * gd[0] - dorig[1] * dorig[2] * gd[1] + (s2 - dorig[2] * dorig[2]) * gd[2]) * inv32;
        }
        dL_dmean3D[3 * i] = gm[0]; dL_dmean3D[3 * i + 1] = gm[1]; dL_dmean3D[3 * i + 2] = gm[2];

        /* ---- cov3D -> scale, rotation (backward.cu:375-438) ---- */
        if (scales) {
            float R[3][3], Mm[3][3], dS[3][3], dM[3][3], G[3][3];
            const float* q = rotations + 4 * i;
            quat_to_rot(q, R);
            const float sv[3] = {scale_modifier * scales[3 * i], scale_modifier * scales[3 * i + 1], scale_modifier * scales[3 * i + 2]};
            for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) Mm[r][cc] = sv[r] * R[cc][r];
            dS[0][0] = gS[0]; dS[1][1] = gS[3]; dS[2][2] = gS[5];
            dS[0][1] = dS[1][0] = 0.5f * gS[1]; dS[0][2] = dS[2][0] = 0.5f * gS[2]; dS[1][2] = dS[2][1] = 0.5f * gS[4];
            for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++)
                dM[r][cc] = 2.0f * (Mm[r][0] * dS[0][cc] + Mm[r][1] * dS[1][cc] + Mm[r][2] * dS[2][cc]);
            for (int r = 0; r < 3; r++)   /* no scale_modifier factor, as in the reference */
                dL_dscale[3 * i + r] = R[0][r] * dM[r][0] + R[1][r] * dM[r][1] + R[2][r] * dM[r][2];
            for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) G[r][cc] = sv[r] * dM[r][cc];
            const float r_ = q[0], x = q[1], y = q[2], z = q[3];
            dL_drot[4 * i + 0] = 2 * z * (G[0][1] - G[1][0]) + 2 * y * (G[2][0] - G[0][2]) + 2 * x * (G[1][2] - G[2][1]);
            dL_drot[4 * i + 1] = 2 * y * (G[1][0] + G[0][1]) + 2 * z * (G[2][0] + G[0][2]) + 2 * r_ * (G[1][2] - G[2][1]) - 4 * x * (G[2][2] + G[1][1]);
            dL_drot[4 * i + 2] = 2 * x * (G[1][0] + G[0][1]) + 2 * r_ * (G[2][0] - G[0][2]) + 2 * z * (G[1][2] + G[2][1]) - 4 * y * (G[2][2] + G[0][0]);
            dL_drot[4 * i + 3] = 2 * r_ * (G[0][1] - G[1][0]) + 2 * x * (G[2][0] + G[0][2]) + 2 * y * (G[1][2] + G[2][1]) - 4 * z * (G[1][1] + G[0][0]);
        }
    }
}

#ifdef _OPENMP
#include <omp.h>
int orc_num_threads(void) { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
int orc_num_threads(void) { return 1; }
void orc_set_num_threads(int n) { (void)n; }
#endif

/* Test hook: SH colour (before +0.5 / clamp) through the same sh_basis() the preprocess uses. */
void orc_eval_sh(int N, int D, int M, const float* dirs, const float* shs /* N x M x 3 */, float* out /* N x 3 */)
{
    for (int i = 0; i < N; i++) {
        float B[16];
        const int nb = sh_basis(D, dirs + 3 * i, B);
        const float* sh = shs + (size_t)i * M * 3;
        for (int ch = 0; ch < 3; ch++) {
            float r = B[0] * sh[ch];
            for (int k = 1; k < nb; k++) r = r + B[k] * sh[3 * k + ch];
            out[3 * i + ch] = r;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * distCUDA2 (submodules/simple-knn/simple_knn.cu:132-183, spatial.cu:15-26): mean of the three
 * smallest squared distances to the other points.  The reference's Morton boxes only prune; the
 * result is the exact 3-NN answer, restated here as a brute-force scan (O(P^2), small P only).
 * ---------------------------------------------------------------------------------------- */
void orc_knn_mean_dist2(int P, const float* pts, float* out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float best[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
        const float* p = pts + 3 * i;
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            const float dx = pts[3 * j] - p[0], dy = pts[3 * j + 1] - p[1], dz = pts[3 * j + 2] - p[2];
            float d = dx * dx + dy * dy + dz * dz;
            for (int k = 0; k < 3; k++) if (best[k] > d) { const float t = best[k]; best[k] = d; d = t; }
        }
        out[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}
