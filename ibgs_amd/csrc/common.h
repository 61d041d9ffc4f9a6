// Shared declarations for the gfx950 plane rasterizer (internal; the public ABI is include/ibgs_rast.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include "../../include/ibgs_rast.h"

namespace ibgs {

constexpr int TILE = IBGS_TILE;
constexpr int WAVE = 64;
constexpr int REC_FLOATS = 16;   // one 64-byte record per Gaussian (see GaussRec)
constexpr int GACC_FLOATS = 16;  // one 64-byte gradient accumulation row per Gaussian

// Field indices of the per-Gaussian render record written by preprocess and staged (as 16-byte
// quads) by the render kernels.  Quad 0 = {x, y, opacity, pad}, quad 1 = {conic a,b,c, dist},
// quad 2 = {r,g,b, pad}, quad 3 = {nx,ny,nz, pad}.
constexpr int IBGS_CULL_WORDS = 4;                         // 64-bit mask words per Gaussian
constexpr int IBGS_CULL_MAX_TILES = 64 * IBGS_CULL_WORDS;  // rectangles up to 256 tiles are culled per tile

enum RecField { R_X = 0, R_Y = 1, R_OP = 2, R_PAD0 = 3, R_CA = 4, R_CB = 5, R_CC = 6, R_DIST = 7,
                R_R = 8, R_G = 9, R_B = 10, R_PAD1 = 11, R_NX = 12, R_NY = 13, R_NZ = 14, R_PAD2 = 15 };

// Columns of the gradient accumulation row (render backward -> preprocess backward): pixel sums of
// moments of q = o*G*dL/dalpha (d = Gaussian centre - pixel, l = conic * d), colour and plane gradients.
enum GaccField { G_SX = 0, G_SY = 1, G_AX = 2, G_AY = 3, G_SXX = 4, G_SXY = 5, G_SYY = 6, G_S0 = 7,
                 G_R = 8, G_G = 9, G_B = 10, G_NX = 11, G_NY = 12, G_NZ = 13, G_DIST = 14, G_PAD = 15 };

struct Carver {   // 128-byte aligned carve-up of a caller-owned arena (or size computation with base==0)
    uintptr_t cur;
    explicit Carver(char* base) : cur(reinterpret_cast<uintptr_t>(base)) {}
    template <typename T> T* take(size_t count) {
        cur = (cur + 127) & ~uintptr_t(127);
        T* p = reinterpret_cast<T*>(cur);
        cur += count * sizeof(T);
        return p;
    }
};

struct GeomState {
    float* rec;            // P x 16
    float* depths;         // P
    float* cov3D;          // P x 6
    uint32_t* tiles;       // P   tiles touched
    uint4* fp;             // P   footprint: (x0 | x1<<16, y0 | y1<<16) = the tile rectangle, tightened by the tile cull, and the first 64 bits of the
                           //     row-major mask of its surviving tiles -- ONE 16-byte gather per Gaussian for the binning stage
    uint64_t* tmask_hi;    // P x (IBGS_CULL_WORDS - 1)  mask bits 64.. (rectangles of 65..IBGS_CULL_MAX_TILES tiles only)
    uint4* fp_sorted;      // P   the footprints in depth order (written by the binning's count pass, read by its place pass)
    uint8_t* clamped;      // P   bit ch set when SH colour channel was clamped
    uint64_t* alive64;     // ceil(P / 64)  lane mask per wave of the preprocess kernel: Gaussians that reach a tile list (what sh_color_kernel evaluates)
    uint32_t* tile_partial; // ceil(P / 64) + IBGS_MAX_VIEWS + 1   tiles touched, summed per wave of the preprocess kernel; the word behind the last wave's is the depth
                           //      sort's error flag in deferred sizing: ONE copy hands the host R (it adds the words up) and the flag right after the sort
    uint32_t* sort_key[2]; // P   depth keys (ping-pong)
    uint32_t* sort_val[2]; // P   Gaussian ids (ping-pong); sort_val[0] ends up depth ordered
    uint32_t* offsets;     // P+5: exclusive scan of the tiles touched (synchronous sizing only); [P] = R, [P+1] = depth sort error flag, [P+2] = C: read back
                           //      together; [P+3] = Gaussians with tiles (what the depth sort keeps); [P+4] = 1 when the depth order lies in sort_val[1]
    uint32_t* hist;        // radix histogram + scan scratch
    size_t hist_elems;
    static GeomState carve(char* base, size_t P, size_t* total);
};

struct ImgState {
    uint32_t* ranges;      // tiles x 2
    float* final_T;        // HW
    uint32_t* n_contrib;   // HW
    float* sum_w;          // HW   (geo) sum of median buffer weights
    uint32_t* low_high;    // HW x 2 (geo) min / max contributor of the median buffer
    int32_t* valid_idx;    // 5 x HW (geo)
    float* valid_w;        // 5 x HW (geo)
    uint32_t* meta;        // 32 words written by the forward for the backward: [0] = buffer_length of a geo pass, [10] = waves per tile of the forward variant that wrote tile_walked, [11] = 1 when the caller's tile_order_hint holds a valid order
    uint32_t* slot_c;      // 8 x HW (geo) contributor number (1-based list position) of every median buffer slot, 0 = empty
    uint32_t* tile_walked; // tiles x 4   how far the forward walked every tile's list (largest n_contrib), per wave of the tile ([tile * waves + wave]) = the backward's work there
    uint32_t* tile_order;  // tiles rounded up to 1024   launch order of the colour backward: workgroup -> tile (render_bwd.hip, balanced placement)
    uint32_t* tile_risky;  // tiles x 4   per wave of the tile: non-zero when the forward staged a record whose conic is near-singular (conic_takes_ref_power) -- the backward walks such
                           //             tiles with its reference-arithmetic kernels (render_bwd.hip)
    static ImgState carve(char* base, int W, int H, size_t* total);
};

constexpr int BIN_CELL = 8;        // two-level binning (binning.hip): a coarse cell = 8 x 8 tiles
constexpr int BIN_XCHUNK = 256;    // coarse entries per expansion chunk (one wave, four rounds)

struct BinState {
    uint32_t* point_list;  // R     sorted Gaussian ids (final); FIRST in the arena so that its offset does not depend on the capacity
    uint4* cent;           // ccap  coarse entries, cell by cell, depth order inside a cell: {Gaussian id, -, its surviving tiles inside the cell as a
                           //       64-bit mask (lo, hi), bit ly * 8 + lx}
    uint32_t* cnt;         // cnt_elems >= ncells x blocks   entries per (cell, block of depth ranks), scanned over the blocks in place
    size_t cnt_elems;
    uint32_t* cell_total;  // ncells   entries per cell
    uint32_t* cell_start;  // ncells + 1   first coarse entry of every cell
    uint32_t* cell_chunk0; // ncells + 1   first expansion chunk of every cell ([ncells] = number of chunks)
    uint32_t* chunk_cnt;   // nchunks_max x 64   per chunk and tile of its cell: count, then prefix within the cell
    uint32_t* tile_total;  // ntiles + 1   entries per tile, scanned in place to the tile starts
    uint32_t* scan_scratch; size_t scan_elems;
    size_t ccap;           // capacity of the coarse arrays (= the capacity the arena was carved for: C <= R always)
    static BinState carve(char* base, size_t R, int W, int H, size_t* total);       // H = height of the (stacked) tile grid in pixels
};

// ---- what the blend kernels stage instead of the record's conic and opacity ------------------------------------------------------
// alpha = o exp(-p2 / 2) = exp2(-(p2 * 0.5 log2 e - log2 o)).  Staging a record into LDS, the blend kernels multiply its conic
// (a, b, c) by 0.5 log2(e) and replace the opacity by -log2(o), ONCE per (Gaussian, tile).  The lane's quadratic form is started from
// that offset (an fma instead of a mul), so what the inner loop holds per pixel is E = -log2(alpha before the 0.99 clamp):
//   alpha      = min(0.99, exp2(-E))             (exp2 with a free source negation: no multiply by o, none by log2 e)
//   skip test  : alpha >= 1/255  <=>  E <= log2(255), ONE compare against a constant (forward.cu:420-425 tests power > 0 and
//                alpha < 1/255; power <= 0 holds for every positive definite conic up to rounding noise; the Gaussians for which that
//                noise can make the reference's own formula positive take the branch described at conic_is_risky below)
// Two multiplies less per (pixel, Gaussian) pair in both passes, which take the same decisions because they stage the same numbers.
// The record keeps the unscaled conic and the opacity (bit-identical to the oracle's; preprocess_bwd reads them).
constexpr float EXP2_SCALE = 0.5f * 1.4426950408889634f;
constexpr float EXP2_UNSCALE = 1.0f / EXP2_SCALE;
constexpr float ALPHA_SKIP_E = 7.9943533f;                    // log2(255): E above this = alpha below 1/255
__device__ __forceinline__ void stage_for_exp2(float4& pos_opacity_quad, float4& conic_quad)
{
    conic_quad.x *= EXP2_SCALE; conic_quad.y *= EXP2_SCALE; conic_quad.z *= EXP2_SCALE;
    pos_opacity_quad.z = -__builtin_amdgcn_logf(pos_opacity_quad.z);          // v_log_f32 = log2; o = 0 -> +inf -> never passes
}

// ---- the reference's `power > 0` skip (forward.cu:420, backward.cu:645) -------------------------------------------------------------
// `power` = -0.5 (a dx^2 + c dy^2) - b dx dy is <= 0 for a positive definite conic; the reference's fp32 evaluation can come out
// positive only through rounding: |rounding| <= ~4 ulp of S = a dx^2 + c dy^2, while -power >= S (1 - |b| / sqrt(a c)) / 2, so a pair can
// be dropped by that test only when b^2 > (1 - 1e-6) a c -- needles hundreds of pixels long and half a pixel wide, up to conics that
// the fp32 inversion of cov2D has left INDEFINITE (then `power > 0` holds on a whole sector of the image and the test is what keeps
// exp(power) from exploding).  conic_is_risky() keeps a decade of margin; it is the same expression as oracle/ibgs_oracle.c's.
// For those Gaussians (a wave-uniform, rare branch: one scalar bit test per staged record otherwise) the blend kernels evaluate
// the reference's own expression per pixel, without contraction -- as the oracle does; nvcc's fma contraction of it is not knowable
// here -- drop the pairs with power > 0 and take E from it, so that they follow the reference's rounding instead of the shifted
// quadratic form's.  They are also exempt from the tile cull (preprocess.hip).  IBGS_FLAG_NO_REF_POWER_SKIP switches the branch off.
constexpr float POWER_RISK = 0.99999f;
constexpr float LOG2_E = 1.4426950408889634f;
__device__ __forceinline__ bool conic_is_risky(float a, float b, float c) { return b * b > POWER_RISK * (a * c); }
// The blend kernels take E from the reference's per-pixel expression for a WIDER class than the one the `power > 0` test can fire for (round 5): the shifted
// quadratic form's rounding error is a few ulp of S = a dx^2 + c dy^2, which stands to E as 1 : (1 - b^2 / (a c)) -- at 10^-3 from singular every ulp of S is
// 10^3 ulp of E.  For conics within 10^-3 of singular both passes therefore follow the reference's rounding (the same wave-uniform branch; such conics are
// rare outside needle scenes: no measurable cost on the bench workloads).  Measured on tests/test_gpu_anisotropic.py's scenes: colour mean L1 against the
// oracle 5.3e-07 .. 3.3e-06 -> 1.8e-08 (giant needles), 1.1e-06 -> 2.2e-07 (needles), n_contrib equal on 99.997 % -> 100 % of the pixels.  The tile cull's
// exemption keeps POWER_RISK: its 0.1 % margin covers this class (4 ulp of S <= 2.4e-4 E here).
constexpr float BLEND_REF_POWER_RISK = 0.999f;
__device__ __forceinline__ bool conic_takes_ref_power(float a, float b, float c) { return b * b > BLEND_REF_POWER_RISK * (a * c); }
__device__ __forceinline__ float ref_power_E(float dx, float dy, float a, float b, float c, float neg_log2_opacity)
{
#pragma clang fp contract(off)
    const float power = -0.5f * (a * dx * dx + c * dy * dy) - b * dx * dy;
    return power > 0.0f ? __builtin_inff() : fmaf(-power, LOG2_E, neg_log2_opacity);          // +inf fails every E <= log2(255) test
}

// ---- exact tile cull, row by row (preprocess.hip counts / masks with it, binning.hip recomputes it for rectangles too large for a mask) ----------
// A Gaussian passes the blend's alpha >= 1/255 test only where q = a dx^2 + 2b dx dy + c dy^2 <= qmax = 2 ln(255 o) (+ 0.1 % + 1e-3).  Inside the
// band of pixel centres of one tile row that ellipse is convex, so the tiles it reaches there are ONE run of consecutive tiles, bounded by the
// ellipse's x-range over the band, whose ends sit at y* = -B sqrt(qmax / (C det)) (upper end) and -y* (lower end) clamped into the band.
// Same operations in the same order as oracle/ibgs_oracle.c (cull_rows_setup / cull_row_run / cull_qmax): basic IEEE operations only and no
// contraction, whatever the translation unit's flags, so that every caller and the oracle get the same runs bit for bit.
__device__ __forceinline__ float ln_portable(float x)
{
#pragma clang fp contract(off)
    uint32_t u = __float_as_uint(x);
    const int e = (int)(u >> 23) - 127;
    u = (u & 0x007FFFFFu) | 0x3F800000u;
    const float m = __uint_as_float(u);
    const float s = (m - 1.0f) / (m + 1.0f), z = s * s;
    const float poly = 1.0f + z * (0.33333334f + z * (0.2f + z * (0.14285715f + z * 0.11111111f)));
    return (float)e * 0.6931472f + 2.0f * s * poly;
}
__device__ __forceinline__ float cull_qmax(float o)
{
#pragma clang fp contract(off)
    return 2.0f * ln_portable(255.0f * o) * 1.001f + 0.001f;
}
struct CullRows { float px, py, B, det, invA, aq, ymax, ystar; int x0, x1; };          // x0, x1: the tightened rectangle's columns
__device__ __forceinline__ void cull_rows_setup(CullRows& j, float px, float py, float A, float B, float C, float det, float qmax, int x0, int x1)
{
#pragma clang fp contract(off)
    j.px = px; j.py = py; j.B = B; j.det = det; j.x0 = x0; j.x1 = x1;
    j.invA = 1.0f / A; j.aq = A * qmax;
    j.ymax = sqrtf(j.aq / det); j.ystar = -B * sqrtf(qmax / (C * det));
}
// ... from the record's conic alone (what the binning and the wave-cooperative walks have at hand).  The determinant is formed HERE, uncontracted: written at
// a call site in a translation unit that contracts (binning.hip) it became fma(A, C, -B B), an ulp away from preprocess's and the oracle's A C - B B, and one
// tile of one row of a 1 386-tile rectangle went missing (tools/fuzz_parity.py ... big, case 4; tests/test_gpu_trained_scene.py keeps the scene).
__device__ __forceinline__ void cull_rows_setup_conic(CullRows& j, float px, float py, float A, float B, float C, float qmax, int x0, int x1)
{
#pragma clang fp contract(off)
    const float det = A * C - B * B;
    cull_rows_setup(j, px, py, A, B, C, det, qmax, x0, x1);
}
// tiles [t0, t1] of tile row ty; false when the row holds none.  (median of (v, lo, hi) == the oracle's v < lo ? lo : (v > hi ? hi : v) for
// lo <= hi and finite v: a pure selection, one v_med3_f32)
__device__ __forceinline__ bool cull_row_run(const CullRows& j, int ty, int& t0, int& t1)
{
#pragma clang fp contract(off)
    const float Y0 = (float)(ty * 16) - j.py, Y1 = Y0 + 15.0f;
    const float yb0 = Y0 > -j.ymax ? Y0 : -j.ymax, yb1 = Y1 < j.ymax ? Y1 : j.ymax;
    if (yb0 > yb1) return false;
    const float yu = __builtin_amdgcn_fmed3f(j.ystar, yb0, yb1), yl = __builtin_amdgcn_fmed3f(-j.ystar, yb0, yb1);
    const float eu = j.aq - j.det * yu * yu, el = j.aq - j.det * yl * yl;
    const float du = eu > 0.0f ? eu : 0.0f, dl = el > 0.0f ? el : 0.0f;
    const float xhi = (-j.B * yu + sqrtf(du)) * j.invA, xlo = (-j.B * yl - sqrtf(dl)) * j.invA;
    t0 = (int)ceilf((xlo - 0.01f + j.px - 15.0f) / 16.0f); t1 = (int)floorf((xhi + 0.01f + j.px) / 16.0f);
    t0 = max(t0, j.x0); t1 = min(t1, j.x1 - 1);
    return t1 >= t0;
}

// ---- diagnostic build only (-DIBGS_TRACE_WAVES, tools/wave_trace.py): where and when every wave of a blend kernel ran ----------------------
// One stamp per workgroup: {HW_ID (wave slot, SIMD, CU, SE), XCC_ID, start, end} on the 100 MHz constant clock, which all XCDs share.  The
// product build compiles none of this (no stamp executes in the real kernel: cdna_hip_programming.md, in-kernel stamps).
#ifdef IBGS_TRACE_WAVES
constexpr int IBGS_TRACE_MAX = 32768;
#define IBGS_TRACE_BEGIN() const unsigned long long ibgs_trace_t0_ = __builtin_amdgcn_s_memrealtime()
#define IBGS_TRACE_END(buf)                                                                                         \
    do {                                                                                                            \
        const unsigned long long t1_ = __builtin_amdgcn_s_memrealtime();                                            \
        if (threadIdx.x == 0 && blockIdx.x < (unsigned)IBGS_TRACE_MAX)                                              \
            (buf)[blockIdx.x] = make_uint4(__builtin_amdgcn_s_getreg((31 << 11) | 4), __builtin_amdgcn_s_getreg((31 << 11) | 20), \
                                           (uint32_t)ibgs_trace_t0_, (uint32_t)t1_);                                \
    } while (0)
#else
#define IBGS_TRACE_BEGIN() do { } while (0)
#define IBGS_TRACE_END(buf) do { } while (0)
#endif

// ---- diagnostic build only (-DIBGS_COUNT_LANES, tools/lane_stats.py): how full are the blend kernels' lanes? ---------------------------------
// Per kernel four 64-bit counters: [0] list entries walked (wave x entry), [1] entries that at least one pixel of the wave blends, [2] lanes that
// EXECUTE a pair evaluation (64 per quadrant evaluation that is not skipped), [3] lanes whose pixel really blends the Gaussian.  Each wave adds
// its totals once, at its end.  The product build compiles none of this.
#ifdef IBGS_COUNT_LANES
#define IBGS_LANES_DECL() unsigned long long ibgs_lc_[4] = {0ull, 0ull, 0ull, 0ull}
#define IBGS_LANES_ADD(i, n) ibgs_lc_[i] += (unsigned long long)(n)
#define IBGS_LANES_FLUSH(buf) do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 4; i_++) atomicAdd(&(buf)[i_], ibgs_lc_[i_]); } } while (0)
#else
#define IBGS_LANES_DECL() do { } while (0)
#define IBGS_LANES_ADD(i, n) do { } while (0)
#define IBGS_LANES_FLUSH(buf) do { } while (0)
#endif

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define IBGS_HIP(expr)                                                                    \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            ::ibgs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -IBGS_ERR_HIP;                                                         \
        }                                                                                 \
    } while (0)

// ---- optional stage timing (api.hip; include/ibgs_rast.h: ibgs_timing_enable) ---------------------------------------------------
struct StageTimer {   // RAII: records an event pair around one stage (or one kernel of it) when that stage is selected
    hipStream_t s; hipEvent_t a = nullptr, b = nullptr; int stage; bool on;
    StageTimer(hipStream_t s_, int stage_);
    ~StageTimer();
};

// ---- host launchers (one per stage) ---------------------------------------------------------
struct Cam {          // camera block handed to kernels by value; matrices stay in device memory
    const float* vm;      // 16, transposed world->view   (wave-uniform reads -> scalar loads)
    const float* pm;      // 16, transposed full projection
    const float* campos;  // 3
    const float* bg;      // 3
    float tanfovx, tanfovy, fx, fy;
    int W, H, gx, gy;
};
inline Cam make_cam(const float* vm, const float* pm, const float* campos, const float* bg,
                    float tanfovx, float tanfovy, int W, int H)
{
    Cam c;
    c.vm = vm; c.pm = pm; c.campos = campos; c.bg = bg;
    c.tanfovx = tanfovx; c.tanfovy = tanfovy;
    c.fy = H / (2.0f * tanfovy); c.fx = W / (2.0f * tanfovx);   // rasterizer_impl.cu:362-363
    c.W = W; c.H = H; c.gx = (W + TILE - 1) / TILE; c.gy = (H + TILE - 1) / TILE;
    return c;
}


// ---- fused plane-map glue ------------------------------------------------------------------
// The reference builds all_map = [n_cam, 1, |d_cam|] with ~10 torch kernels per call (gaussian_renderer/__init__.py:
// 304-316; scene/gaussian_model.py:156-173).  With plane_mode != 0 the preprocess kernels do it per Gaussian:
//   mode 1 (learnt normal):  n = raw / |raw|, offset = raw offset            (get_normal)
//   mode 2 (smallest axis):  n = column argmin(scale) of R(rotation)          (get_normal_w_smallest_axis)
//   flip n (and the offset) towards the camera; n_cam = n @ V[:3,:3]; d = |-(n . x) + offset - n_cam . V[3,:3]|
// V = world_view_transform, i.e. the 16 floats of cam.vm read row-major.
struct PlaneEval {
    float n[3];        // world normal after the flip
    float ncam[3];     // camera-frame normal (all_map[0..2])
    float dist;        // all_map[4]
    float flip;        // +1 / -1
    float sgn;         // sign of the signed camera-frame distance (derivative of the abs)
    float inv_len;     // mode 1: 1 / |raw|
    int axis;          // mode 2: column index
};

__device__ __forceinline__ PlaneEval plane_eval(int mode, const float* __restrict__ raw_n, const float* __restrict__ raw_off,
                                                const float* __restrict__ scales, const float* __restrict__ rot, int i,
                                                float mx, float my, float mz, const float* __restrict__ campos,
                                                const float* __restrict__ vm)
{
    PlaneEval e;
    float off = 0.f;
    e.inv_len = 1.f; e.axis = 0;
    if (mode == IBGS_PLANE_LEARNT) {
        const float a = raw_n[3 * i], b = raw_n[3 * i + 1], c = raw_n[3 * i + 2];
        const float len = sqrtf(a * a + b * b + c * c);
        e.inv_len = 1.0f / len;
        e.n[0] = a / len; e.n[1] = b / len; e.n[2] = c / len;
        off = raw_off ? raw_off[i] : 0.f;
    } else {
        const float s0 = scales[3 * i], s1 = scales[3 * i + 1], s2 = scales[3 * i + 2];
        int k = 0; float sm = s0;
        if (s1 < sm) { sm = s1; k = 1; }
        if (s2 < sm) { sm = s2; k = 2; }
        e.axis = k;
        const float r = rot[4 * i], x = rot[4 * i + 1], y = rot[4 * i + 2], z = rot[4 * i + 3];
        if (k == 0)      { e.n[0] = 1.f - 2.f * (y * y + z * z); e.n[1] = 2.f * (x * y + r * z);       e.n[2] = 2.f * (x * z - r * y); }
        else if (k == 1) { e.n[0] = 2.f * (x * y - r * z);       e.n[1] = 1.f - 2.f * (x * x + z * z); e.n[2] = 2.f * (y * z + r * x); }
        else             { e.n[0] = 2.f * (x * z + r * y);       e.n[1] = 2.f * (y * z - r * x);       e.n[2] = 1.f - 2.f * (x * x + y * y); }
    }
    const float tc = e.n[0] * (campos[0] - mx) + e.n[1] * (campos[1] - my) + e.n[2] * (campos[2] - mz);
    e.flip = (tc < 0.0f) ? -1.f : 1.f;
    e.n[0] *= e.flip; e.n[1] *= e.flip; e.n[2] *= e.flip; off *= e.flip;
#pragma unroll
    for (int j = 0; j < 3; j++) e.ncam[j] = e.n[0] * vm[j] + e.n[1] * vm[4 + j] + e.n[2] * vm[8 + j];
    float gd = -(e.n[0] * mx + e.n[1] * my + e.n[2] * mz);
    if (mode == IBGS_PLANE_LEARNT) gd += off;
    const float sd = gd - (e.ncam[0] * vm[12] + e.ncam[1] * vm[13] + e.ncam[2] * vm[14]);
    e.sgn = sd > 0.f ? 1.f : (sd < 0.f ? -1.f : 0.f);
    e.dist = fabsf(sd);
    return e;
}

// ---- workgroup -> work item (tile, wave of the tile) ------------------------------------------------------------------------------
// Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own 4 MB L2 (MI355X_MICROARCH.md).  Which tiles share
// an XCD decides (i) whether the two halves of a 128-byte line of an output plane -- 32 pixels = two tiles side by side -- meet in ONE
// L2 and leave as a full line, (ii) how much of the source textures' halo the geo epilogue's gathers find already cached, (iii) how
// evenly a non-uniform image spreads over the chip.  Three layouts, chosen per kernel by measurement (docs/EXPERIMENTS.md section 7):
//   RR     item = workgroup id: neighbouring items on different XCDs (best balance, every output line split over two L2s)
//   GROUP  runs of `g` consecutive items per XCD: with g x (waves per tile) covering 2+ tiles, output lines are completed in one L2
//   BLOCK  bx x by tile blocks per XCD (2-D locality for the gathers)
// (Round 6: the layouts are fixed in the launchers -- 8 x 8 blocks, 8 x 4 for the geo backward; the environment overrides of rounds 3-5 are gone with their sweeps.)
enum { TMAP_RR = 0, TMAP_GROUP = 1, TMAP_BLOCK = 2 };
struct TileMap { int mode, g, bx, by; };
inline int tile_map_grid(const TileMap& m, int gx, int gy, int ipt)
{
    if (m.mode == TMAP_BLOCK) {
        const int nb = ((gx + m.bx - 1) / m.bx) * ((gy + m.by - 1) / m.by);
        return ((nb + 7) / 8) * 8 * m.bx * m.by * ipt;
    }
    const int g = m.mode == TMAP_GROUP ? m.g : 1;
    return ((gx * gy * ipt + 8 * g - 1) / (8 * g)) * 8 * g;
}
__device__ __forceinline__ bool tile_map_item(const TileMap& m, int b, int gx, int gy, int ipt, int& tile, int& sub)
{
    const int xcd = b & 7, idx = b >> 3;
    if (m.mode == TMAP_BLOCK) {
        const int per = m.bx * m.by * ipt;
        const int blk = (idx / per) * 8 + xcd, within = idx % per;
        const int nbx = (gx + m.bx - 1) / m.bx;
        const int t = within / ipt;
        const int tx = (blk % nbx) * m.bx + t % m.bx, ty = (blk / nbx) * m.by + t / m.bx;
        sub = within % ipt;
        tile = ty * gx + tx;
        return tx < gx && ty < gy;
    }
    int item = b;
    if (m.mode == TMAP_GROUP) item = ((idx / m.g) * 8 + xcd) * m.g + idx % m.g;
    tile = item / ipt; sub = item % ipt;
    return item < gx * gy * ipt;
}

// hybrid colour kernels (render_fwd.hip, render_bwd.hip): a tile is walked by four quadrant waves instead of one tile wave when its work exceeds
// theta per cent of a SIMD's fair share of the frame's (total / 1024)
__device__ __forceinline__ bool hybrid_split(uint32_t work, unsigned long long total, uint32_t theta_pct)
{
    return (unsigned long long)work * 1024ull * 100ull > total * theta_pct;
}
// A launch order (ImgState::tile_order, ibgs_backward_args::tile_order_out, ibgs_forward_args::tile_order_hint) holds one word per slot: the tile, or
// 0xFFFFFFFF for an empty slot; bit 31 of a tile's word: the backward that wrote the order found the tile heavy enough for four quadrant waves.
constexpr uint32_t ORDER_SPLIT_BIT = 0x80000000u;
// workgroup -> (tile, wave of the tile, split?) for the hybrid kernels under a launch order of grid1 slots: the first grid1 workgroups are wave 0 of the slots'
// tiles, the 3 x grid1 behind them waves 1..3.  A tile's four side by side would put the work of a frame of tile waves
// into every fourth workgroup, and the dispatcher deals consecutive workgroups round-robin: a quarter of the SIMDs got all of it (3.0 against 1.24 ms at 720p).
__device__ __forceinline__ bool hybrid_item(int b, int grid1, const uint32_t* __restrict__ order, int& tile, int& sub, bool& split)
{
    // (workgroup b runs on XCD b % 8, and grid1 is a multiple of 8: waves 1..3 of a slot land on the XCD of its wave 0 -- one L2 fetches the tile's records)
    const int idx = (b - grid1) >> 3;
    const int slot = b < grid1 ? b : (idx / 3) * 8 + (b & 7);
    sub = b < grid1 ? 0 : idx % 3 + 1;
    const uint32_t t = order[slot];
    split = (t & ORDER_SPLIT_BIT) != 0u;
    tile = (int)(t & ~ORDER_SPLIT_BIT);
    return t != 0xFFFFFFFFu;
}
// ... and without one: the tiles in the tile map's order (grid1 = tile_map_grid(m, gx, gy, 1), then tile_map_grid(m, gx, gy, 3) workgroups)
__device__ __forceinline__ bool hybrid_item_mapped(const TileMap& m, int b, int grid1, int gx, int gy, int& tile, int& sub)
{
    if (b < grid1) { const bool ok = tile_map_item(m, b, gx, gy, 1, tile, sub); sub = 0; return ok; }
    const bool ok = tile_map_item(m, b - grid1, gx, gy, 3, tile, sub);
    sub += 1;
    return ok;
}
constexpr int HYBRID_THETA = 50;
// Frames of fewer than HYBRID_MIN_TILES tiles: a SIMD's fair share is less than a tile, nearly every tile would be split -- plain quadrant waves, without the
// order kernel in front (400 x 400 / 10 k Gaussians, BASELINE's C1: backward 0.29 -> 0.20 ms).
constexpr int HYBRID_MIN_TILES = 768;
// colour passes of frames with fewer tiles than this use the hybrid kernels
constexpr int hybrid_max_tiles() { return 4096; }
constexpr int hybrid_theta() { return HYBRID_THETA; }

// The note the host waits for when it sizes the tile lists (api.hip, ibgs_forward): the per-wave tile sums of the preprocess kernel added up by ONE workgroup
// of NT threads, the total and then a ticket stored straight into pinned, host-coherent memory.  Either rendered_note_kernel (api.hip) or -- round 6, a launch
// fewer -- workgroup 0 of the SH colour kernel on its way in (preprocess.hip: the kernel behind the geometry kernel in the stream anyway).
struct RenderedNote { const uint32_t* partial; uint32_t nwords; uint32_t ticket; uint32_t* host; };
template <int NT>
__device__ __forceinline__ void rendered_note_block(const RenderedNote& n, unsigned long long* s_w /* NT / 64 words of LDS */)
{
    unsigned long long sum = 0;
    for (uint32_t i = threadIdx.x; i < n.nwords; i += NT) sum += n.partial[i];
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < NT / 64; w++) t += s_w[w];
        volatile uint32_t* host_words = n.host;
        host_words[0] = (uint32_t)t; host_words[1] = (uint32_t)(t >> 32);
        __threadfence_system();                                   // the total is visible to the host before the ticket is
        __hip_atomic_store(n.host + 2, n.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// phase 0: everything; 1: the geometry kernel(s) alone; 2: what phase 1 left out (the SH colours).  note (phase 2): carried by the SH kernel's first workgroup when
// such a kernel is launched -- the return value is then 1 (0: the caller launches rendered_note_kernel itself; < 0: error)
int launch_preprocess(hipStream_t s, const ibgs_forward_args& a, const GeomState& g, int phase = 0, const RenderedNote* note = nullptr);
int launch_mark_visible(hipStream_t s, int P, const float* means3D, const float* vm, uint8_t* present);

// device-wide primitives (scan_sort.hip)
size_t radix_hist_elems(size_t n);      // scratch (uint32 elements) needed by radix_sort_pairs on n items
// Stable LSD radix sort of (key,val) pairs on key bits [0, nbits). Result lands in keys[0]/vals[0].
int radix_sort_pairs(hipStream_t s, uint32_t* keys[2], uint32_t* vals[2], size_t n, int nbits,
                     uint32_t* hist, size_t hist_elems, uint32_t* err_dev = nullptr, uint32_t* kept_dev = nullptr, bool scratch_is_zero = false,
                     uint32_t* result_alt = nullptr);
// result_alt: device word (zeroed by the caller).  When given (32-bit keys, single-launch passes), a LAST pass in which every key carries the same digit --
//          the top byte of depths within [2, 8), say -- moves nothing and sets *result_alt = 1: the result is then in keys[1] / vals[1] (else, as always, in [0])
size_t radix_zero_elems(size_t n, int nbits);      // leading words of `hist` the sort needs zeroed (see scratch_is_zero)
// err_dev: device word (zeroed by the caller) that the single-launch look-back passes set to 1 when their bounded spin gives up --
//          the pass has then scattered with a partial prefix; the caller must read it back and fail the call.  With err_dev == nullptr
//          radix_sort_pairs reads its own flag back (one stream synchronisation) and fails the call itself
// kept_dev: device word (zeroed by the caller).  When given, pairs whose key is 0xFFFFFFFF need not be carried: *kept_dev receives the
//          number K of other pairs, the result holds those K pairs sorted in [0, K) and unspecified pairs behind them
// scratch_is_zero: the caller has zeroed hist[0, radix_zero_elems(n, nbits)) on the stream already
// Exclusive scan of `n` uint32 (in place allowed); out[n] receives the total when with_total.
int exclusive_scan_u32(hipStream_t s, const uint32_t* in, uint32_t* out, size_t n, uint32_t* scratch,
                       size_t scratch_elems, bool with_total);
size_t scan_scratch_elems(size_t n);
void radix_set_onesweep(bool on);       // default on; off = hist + scan + scatter launches per pass
void radix_set_lookback_spins(uint32_t v);      // tests: how often a look-back sleeps on an unpublished word before its pass gives up (default 2^26)

// per-tile lists + tile ranges from the depth-ordered Gaussians (two-level binning, binning.hip); `cap` = capacity of point_list
// part 1 (ranges + counters; returns the sort buffer index >= 0, or an error < 0) and part 2 (the lists themselves)
int launch_binning(hipStream_t s, int P, int64_t cap, int gx, int gy, const GeomState& g, const BinState& b, uint32_t* ranges,
                   const uint32_t* order_hint = nullptr, uint32_t* meta = nullptr /* meta[11] = 1 when the hint is a valid tile order */,
                   int n_views = 1 /* batched depth views: P = n_views x instances, gy = n_views x rows */,
                   const uint32_t* sort_flag = nullptr, uint32_t* host_note = nullptr /* pinned host words the last kernel leaves R, C and the sort's error word in */,
                   uint32_t note_ticket = 0 /* ... and then, behind a system-scope fence, this ticket (host_note[4]): the forward's binning has run */);
constexpr int ORDER_CLASSES = 1024;          // SIMDs of the chip = classes of the balanced launch order (render_bwd.hip)
int launch_binning_scatter(hipStream_t s, int64_t cap, int gx, int gy, const BinState& b);

int launch_pack_rgba(hipStream_t s, const float* src, float4* dst, int W, int H, int n);
// geo backward: per-pixel table of the median / warp terms of every buffered contributor (render_bwd.hip), 6 words per slot
constexpr int GEO_TAB_FIELDS = 6;
inline size_t geo_table_floats(int W, int H) { return (size_t)W * H * IBGS_MAX_BUFFER_LENGTH * GEO_TAB_FIELDS; }
int launch_render_forward(hipStream_t s, const ibgs_forward_args& a, const GeomState& g, const BinState& b,
                          const ImgState& im, const float4* src_rgba);
int launch_render_backward(hipStream_t s, const ibgs_backward_args& a, const GeomState& g, const BinState& b,
                           const ImgState& im, const float4* src_rgba, float* slab = nullptr, float* geo_tab = nullptr);
int launch_preprocess_backward(hipStream_t s, const ibgs_backward_args& a, const GeomState& g);

// IBGS_FLAG_DETERMINISTIC (deterministic.hip): slab of per-(Gaussian, tile) sums + its sort scratch, carved from det_scratch
struct DetState {
    float* slab;           // (R x ipt) x 16, row = position in the sorted list x waves per tile + wave of the tile
    uint32_t* keys[2];     // R   Gaussian ids (ping-pong)
    uint32_t* vals[2];     // R   list positions (ping-pong)
    uint32_t* seg;         // P+1 first sorted slot of every Gaussian
    uint32_t* hist; size_t hist_elems;
    static DetState carve(char* base, size_t rows, size_t P, size_t* total);          // rows = R x waves per tile
};
int render_backward_waves_per_tile(const ibgs_backward_args& a);
int render_backward_ref_arith(uint32_t flags);          // how the blend backward sums the pairs of near-singular conics (render_bwd.hip: RA_*); bit 4 (RA_ASSOC): their rows hold the reference's sums
int launch_det_prepare(hipStream_t s, const DetState& d, size_t rows);                    // zero the slab
int launch_det_reduce(hipStream_t s, const DetState& d, const uint32_t* point_list, size_t R, int ipt, int P, float* gacc, const uint32_t* listed /* device word: entries the lists hold (<= R) */);
int launch_sh_grad_from_views(hipStream_t s, int P, int D, int M, int n_views, const float* means3D, const float* camposes,
                              const float* dcolor, size_t view_stride, float* dL_dsh);

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack_rect(int lo, int hi) { return (uint32_t)lo | ((uint32_t)hi << 16); }

}  // namespace ibgs
