// F1-F3 + T1: front-to-back alpha blend per tile, plane / median-depth buffer, source-view warp.
//
// Behaviour: DPR/cuda_rasterizer/forward.cu:303-665 (renderCUDA) and the texture set-up of
// rasterizer_impl.cu:34-133.  Organisation is gfx950-native rather than the reference's
// "256-thread block, one thread per pixel, __syncthreads per round":
//
//   * ONE WAVE owns a whole 16x16 tile (PPL = 4: lane l blends pixel (l%8, l/8) of each of the four
//     8x8 quadrants) or one 8x8 quadrant (PPL = 1, used by the register-heavy geo variant).  There
//     is no cross-wave synchronisation at all; a workgroup is a single wave, so "barriers" are only
//     waitcnts and tiles finish independently.
//   * Gaussian records (64 B, written by preprocess) are staged 64 at a time into LDS with 16-byte
//     per-lane loads and read back as wave-uniform broadcasts (ds_read_b128), i.e. 3 LDS reads per
//     Gaussian per 256 pixels.
//   * Each quadrant is skipped with a wave-uniform branch when no lane passes the alpha test (ballot), which
//     recovers 8x8 sub-tile culling.  What a lane holds per pixel is E = -log2(o G) (conic staged in exp2 units, the quadratic form
//     started from -log2 o, common.h); the test is the one compare E <= log2(255): no exp for pairs that fail, and alpha is
//     min(0.99, exp2(-E)) without a multiply.  "Pixel still blending" is a
//     64-bit lane mask kept in SGPRs (one per quadrant): masks are combined on the scalar unit and
//     turned back into predicates with inverse_ballot, so the blend itself is branch-free VALU code and
//     a finished quadrant costs one scalar compare per Gaussian.
//   * Early termination is per wave: the tile stops fetching as soon as every pixel is done.
//   * CUDA layered textures do not exist on gfx950; source images are packed to RGBA float4 once per
//     call and sampled with explicit bilinear gathers that follow the texture unit's addressing rules
//     (unnormalised, clamp, linear).
#include "common.h"

namespace ibgs {

struct FwdParams {
    const uint32_t* ranges; const uint32_t* point_list; const float4* rec;
    Cam cam;
    int ntiles;
    TileMap tmap;
    // batched depth-only views (stacked tile grid): rows per view, per-view focal lengths
    int n_views, gyv; float fxv[IBGS_MAX_VIEWS], fyv[IBGS_MAX_VIEWS];
    // geo
    int n_src; int L; float thr; int tex_quant;
    int power_skip;      // reproduce the reference's `power > 0` skip for near-singular conics (common.h)
    const float* ref_to_src; const float* src_cam_pos; const float4* src_rgba; const float* src_depths;
    int ds0, ds1, ds2, ds3, ds4;          // plane of src_depths that source 0..4 reads (IBGS_FLAG_SRC_DEPTH_SLOTS; else 0..4) -- scalars, not an array: a run-time index must stay a select chain
    // per-pixel state
    float* final_T; uint32_t* n_contrib; float* sum_w; uint32_t* low_high; int32_t* valid_idx; float* valid_w; uint32_t* slot_c; uint32_t* meta; uint32_t* walked;
    uint32_t* risky;          // ImgState::tile_risky (4 words per tile): this wave staged a record whose conic is near-singular
    int hybrid_grid1;    // ... its first hybrid_grid1 workgroups are the tiles' first waves (hybrid_item, common.h)
    int hybrid;          // 1: render_fwd_color_hybrid_kernel
    const uint32_t* order;      // the caller's launch order hint (colour, one wave per tile), checked by an extra workgroup of cell_place_kernel: meta[11]; nullptr: the tile map
    // outputs
    float* out_color; float* out_normal; float* out_depth; float* out_cam_feat; float* out_warped;
    float* out_min_depth_diff; float* out_camera_ray; int32_t* out_mask;
};

enum { MODE_COLOR = 0, MODE_GEO = 1, MODE_DEPTH = 2 };

__global__ void __launch_bounds__(256) pack_rgba_kernel(const float* __restrict__ src, float4* __restrict__ dst, size_t HW, int n)
{   // packRGBA, rasterizer_impl.cu:34-56
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= HW * (size_t)n) return;
    const size_t layer = idx / HW, pix = idx % HW;
    const float* b = src + layer * 3 * HW;
    dst[idx] = make_float4(b[pix], b[HW + pix], b[2 * HW + pix], 1.0f);
}

int launch_pack_rgba(hipStream_t s, const float* src, float4* dst, int W, int H, int n)
{
    const size_t total = (size_t)W * H * n;
    hipLaunchKernelGGL(pack_rgba_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, (size_t)W * H, n);
    IBGS_HIP(hipGetLastError());
    return 0;
}

__device__ __forceinline__ float quant8(float a, int quant) { return quant ? floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f) : a; }

// Linear-filtered, clamp-addressed fetch at unnormalised texture coordinate (x, y) -- texel centres
// at i + 0.5 (what tex2DLayered does with the descriptor of rasterizer_impl.cu:120-126).
__device__ __forceinline__ float4 tex_rgba(const float4* __restrict__ img, int W, int H, float x, float y, int quant)
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fxi = floorf(xb), fyi = floorf(yb);
    const float a = quant8(xb - fxi, quant), b = quant8(yb - fyi, quant);
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

__device__ __forceinline__ float tex_depth(const float* __restrict__ img, int W, int H, float x, float y, int quant)
{
#pragma clang fp contract(off)          // its result is thresholded (source validity): operation by operation like the oracle's tex_1
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fxi = floorf(xb), fyi = floorf(yb);
    const float a = quant8(xb - fxi, quant), b = quant8(yb - fyi, quant);
    const int i0 = min(W - 1, max(0, (int)fxi)), i1 = min(W - 1, max(0, (int)fxi + 1));
    const int j0 = min(H - 1, max(0, (int)fyi)), j1 = min(H - 1, max(0, (int)fyi + 1));
    const float t00 = img[(size_t)j0 * W + i0], t10 = img[(size_t)j0 * W + i1];
    const float t01 = img[(size_t)j1 * W + i0], t11 = img[(size_t)j1 * W + i1];
    return (1.f - a) * (1.f - b) * t00 + a * (1.f - b) * t10 + (1.f - a) * b * t01 + a * b * t11;
}

// Workgroup -> (tile, wave of the tile): tile_map_item, common.h.

// MAXL = compile-time capacity of the per-pixel median buffer (4 covers the reference's default
// buffer_length = 4 with half the select chains of 8).
#ifdef IBGS_TRACE_WAVES
__device__ uint4 g_trace_fwd[IBGS_TRACE_MAX];
#endif
#ifdef IBGS_COUNT_LANES
__device__ unsigned long long g_lanes_fwd[4];
#endif
// The wave's work: `sub` of the 4 / PPL waves of `tile`.  s_rec: the caller's staging area (NQ x CHUNK float4; a kernel that holds two
// instantiations of the body -- the hybrid colour kernel below -- hands both the same one).
// TILE_Q >= 0 (PPL == 1, the hybrid kernel's quadrant waves; = sub, at compile time): the exponent E of the wave's pixels is formed exactly as a tile wave
// forms it for that quadrant -- for the lane's quadrant-0 pixel, then shifted -- so that a pixel's colour does not depend, not by one bit, on which shape
// walked its tile.  (With the quadrant as a run-time value the select between the four expressions became a chain of six scalar branches per list entry:
// quadrant waves 0.35 -> 0.51 ms at 400 x 400.)
template <int MODE, int PPL, int MAXL, int TILE_Q = -1>
__device__ __forceinline__ void render_fwd_body(const FwdParams& p, const int tile, const int sub,
                                                float4 (&s_rec)[(MODE == MODE_GEO) ? 4 : 3][(MODE == MODE_GEO) ? 16 : WAVE])
{
    IBGS_TRACE_BEGIN();
    IBGS_LANES_DECL();
    constexpr bool GEO = (MODE == MODE_GEO);
    constexpr bool DEPTH = (MODE == MODE_DEPTH);
    constexpr int NQ = GEO ? 4 : 3;          // record quads staged per Gaussian
    // Records per staging round.  Colour / depth-only: 64, one record per lane (3 KB: eight waves per SIMD fit).  Geo: 16, lane l fetches quad
    // l % 4 of record l / 4 (one coalesced 64-byte line per record, 1 KB of LDS): with the 4 KB of ring-buffer columns below a wave then
    // holds 5 KB instead of 8, and LDS stops limiting the kernel to five waves per SIMD.  The next round's quad is fetched before the
    // current round is blended (one float4 per lane in flight), so the shorter rounds do not expose their load latency.
    constexpr int CHUNK = GEO ? 16 : WAVE;
    static_assert(sizeof(s_rec) == sizeof(float4) * NQ * CHUNK, "staging area");

    const int lane = threadIdx.x;
    constexpr int IPT = 4 / PPL;                          // work items (waves) per tile: 1, 2 (half tiles: quadrant pairs 0-1 / 2-3) or 4
    const int quad0 = sub * PPL;
    const int W = p.cam.W, H = p.cam.H;
    int trow = tile / p.cam.gx, view = 0;
    if (DEPTH && p.n_views > 1) { view = trow / p.gyv; trow -= view * p.gyv; }      // which camera's grid this tile belongs to
    const int tx0 = (tile % p.cam.gx) * TILE, ty0 = trow * TILE;
    const size_t HW = (size_t)W * H;
    const size_t vbase = (size_t)view * HW;          // offset of this view's planes (0 for a single view)

    int px[PPL], py[PPL];
    float pxf[PPL], pyf[PPL];
    bool inside[PPL];
    uint64_t live[PPL];          // wave-uniform lane masks (SGPR pairs): pixels of quadrant q still blending
    uint64_t recm[PPL];          // geo: ... still feeding their median buffer
    float T[PPL], C[PPL][3];
    uint32_t lastc[PPL];
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        const int qq = quad0 + q;
        px[q] = tx0 + (qq & 1) * 8 + (lane & 7);
        py[q] = ty0 + (qq >> 1) * 8 + (lane >> 3);
        pxf[q] = (float)px[q]; pyf[q] = (float)py[q];
        inside[q] = px[q] < W && py[q] < H;
        live[q] = __builtin_amdgcn_ballot_w64(inside[q]);
        recm[q] = live[q];
        T[q] = 1.0f; C[q][0] = C[q][1] = C[q][2] = 0.f; lastc[q] = 0;
    }
    const float pxf_t0 = (float)(tx0 + (lane & 7)), pyf_t0 = (float)(ty0 + (lane >> 3));          // the lane's pixel in quadrant 0 of the tile (TILE_Q)
    const float fx = (DEPTH && p.n_views > 1) ? p.fxv[view] : p.cam.fx, fy = (DEPTH && p.n_views > 1) ? p.fyv[view] : p.cam.fy;
    const float cx = (float)(W * 0.5f), cy = (float)(H * 0.5f);
    const float eps = 1.0e-8f;

    // geo / depth-only state (PPL == 1 for GEO; DEPTH keeps running sums only plus the ring)
    float Nacc[PPL][3];
    // Median ring buffer (weight, contributor position [, depth]).  The geo pass (one pixel per lane) keeps it in a per-lane
    // LDS column: a slot is then written / read with ONE indexed LDS access instead of MAXL compare-and-select rounds over
    // registers.  The depth-only pass (four pixels per lane, 8 waves per SIMD) cannot afford 12 KB of LDS and keeps registers.
    constexpr bool RING_LDS = GEO && PPL <= 2;
    __shared__ float s_bw[RING_LDS ? PPL : 1][RING_LDS ? MAXL : 1][RING_LDS ? WAVE : 1];
    __shared__ uint32_t s_bc[RING_LDS ? PPL : 1][RING_LDS ? MAXL : 1][RING_LDS ? WAVE : 1];
    float bd[PPL][RING_LDS ? 1 : MAXL], bw[PPL][RING_LDS ? 1 : MAXL]; uint32_t bc[PPL][RING_LDS ? 1 : MAXL];
    int before_ptr[PPL], below_count[PPL];
    float tot_w[PPL], wd_sum[PPL];
    int resume[PPL]; uint32_t cnt[PPL];
    float rayx[PPL], rayy[PPL];
    const int L = p.L;
    if (GEO && blockIdx.x == 0 && lane == 0) p.meta[0] = (uint32_t)L;      // the backward's window pass reads the slot count from here
    const int before_cap = (L % 2 == 0) ? (L / 2) : ((L + 1) / 2);
    const int below_cap = L - before_cap;
    if (GEO || DEPTH) {
#pragma unroll
        for (int q = 0; q < PPL; q++) {
            Nacc[q][0] = Nacc[q][1] = Nacc[q][2] = 0.f;
#pragma unroll
            for (int s = 0; s < MAXL; s++) {
                if (RING_LDS) { s_bw[q][s][lane] = 0.f; s_bc[q][s][lane] = 0u; }
                else if (s < (RING_LDS ? 1 : MAXL)) { bd[q][s] = 0.f; bw[q][s] = 0.f; bc[q][s] = 0u; }
            }
            before_ptr[q] = 0; below_count[q] = 0; tot_w[q] = 0.f; wd_sum[q] = 0.f; resume[q] = 0; cnt[q] = 0;
            rayx[q] = (pxf[q] - cx) / fx; rayy[q] = (pyf[q] - cy) / fy;
        }
    }

    const uint32_t r0 = p.ranges[2 * tile], r1 = p.ranges[2 * tile + 1];
    const int n = (int)(r1 - r0);

    // geo: what lane l stages -- quad l % 4 of entry base + l / 4 (fetched one round ahead)
    auto fetch_quad = [&](int base) -> float4 {
        const int e = base + (lane >> 2);
        if (e < n) return p.rec[(size_t)p.point_list[r0 + e] * 4 + (lane & 3)];
        return make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float4 ahead = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (GEO) ahead = fetch_quad(0);
    int base = 0;          // (lives past the loop: where the wave left the list)
    uint64_t riskany = 0ull;          // a staged record had a near-singular conic: the backward walks this tile with its reference-arithmetic kernel (render_bwd.hip)
    for (; base < n; base += CHUNK) {
        uint64_t riskm = 0ull;          // staged records whose conic is near-singular: the reference's power expression decides for them (common.h)
        if constexpr (GEO) {
            float4 v = ahead;
            const int qd = lane & 3;
            const bool risky = qd == 1 && base + (lane >> 2) < n && conic_takes_ref_power(v.x, v.y, v.z);
            // stage_for_exp2 (common.h), one quad per lane: quad 0 carries the opacity, quad 1 the conic
            if (qd == 0) v.z = -__builtin_amdgcn_logf(v.z);
            if (qd == 1) { v.x *= EXP2_SCALE; v.y *= EXP2_SCALE; v.z *= EXP2_SCALE; }
            s_rec[qd][lane >> 2] = v;
            if (p.power_skip) riskm = __builtin_amdgcn_ballot_w64(risky);          // record j of the round = bit 4 j + 1 (its quad-1 lane)
            ahead = fetch_quad(base + CHUNK);
        } else {   // stage up to 64 records: lane e loads the quads of entry base+e
            const int e = base + lane;
            bool risky = false;
            if (e < n) {
                const uint32_t id = p.point_list[r0 + e];
                const float4* r = p.rec + (size_t)id * 4;
                float4 c0 = r[0], c1 = r[1];
                risky = conic_takes_ref_power(c1.x, c1.y, c1.z);
                stage_for_exp2(c0, c1);                        // conic in units of the exp2 exponent, opacity as -log2 (common.h)
                s_rec[0][lane] = c0;
                s_rec[1][lane] = c1;
                s_rec[2][lane] = DEPTH ? r[3] : r[2];          // rgb (colour / geo) or the normal (depth-only)
            }
            if (p.power_skip) riskm = __builtin_amdgcn_ballot_w64(risky);
        }
        riskany |= riskm;
        __syncthreads();
        const int count = min(CHUNK, n - base);
        for (int j = 0; j < count; j++) {
            const float4 q0 = s_rec[0][j];      // x, y, opacity
            const float4 q1 = s_rec[1][j];      // conic a, b, c, plane distance
            const float4 q2 = s_rec[2][j];      // rgb (colour/geo) or normal (depth-only)
            float4 q3 = q2;
            if constexpr (GEO) q3 = s_rec[3][j];          // normal
            const int e = base + j;
            IBGS_LANES_ADD(0, 1);
#ifdef IBGS_COUNT_LANES
            bool lc_any_ = false;
#endif
            // 1-based list position as ONE vector register per Gaussian (opaque to the optimiser, which otherwise re-materialises
            // the scalar -> vector move inside every quadrant's branch)
            uint32_t e1v = (uint32_t)(e + 1);
            asm volatile("" : "+v"(e1v));
            // p2 = d^T conic d = -2 * power.  With 4 pixels per lane the quadratic form is evaluated once for the
            // lane's pixel in quadrant 0 and shifted to the other three (pixel offsets (8,0), (0,8), (8,8)):
            // p2(d - s) = p2(d) - 2 s^T conic d + s^T conic s -- 14 VALU ops for four pixels instead of 32.
            const float dx0 = q0.x - (TILE_Q >= 0 ? pxf_t0 : pxf[0]), dy0 = q0.y - (TILE_Q >= 0 ? pyf_t0 : pyf[0]);
            const float lx0 = q1.x * dx0 + q1.y * dy0, ly0 = q1.y * dx0 + q1.z * dy0;
            const float P0 = fmaf(dx0, lx0, fmaf(dy0, ly0, q0.z));          // E = -log2(o G) of the lane's quadrant-0 pixel (common.h)
            float p2q[PPL];
            p2q[0] = P0;
            if (PPL >= 2) p2q[1] = fmaf(-16.0f, lx0, P0 + 64.0f * q1.x);
            if (PPL == 4) {
                p2q[2] = fmaf(-16.0f, ly0, P0 + 64.0f * q1.z);
                p2q[3] = fmaf(128.0f, q1.y, p2q[2] + (p2q[1] - P0));          // E(d - (8,8)) = E2 + (E1 - E0) + 128 b: three instructions instead of four
            }
            if constexpr (TILE_Q >= 1 && PPL == 1) {          // the tile wave's expressions for this quadrant, operation by operation
                const float E1 = fmaf(-16.0f, lx0, P0 + 64.0f * q1.x);
                const float E2 = fmaf(-16.0f, ly0, P0 + 64.0f * q1.z);
                p2q[0] = TILE_Q == 1 ? E1 : (TILE_Q == 2 ? E2 : fmaf(128.0f, q1.y, E2 + (E1 - P0)));
            }
            if ((riskm >> (GEO ? 4 * j + 1 : j)) & 1ull) {          // wave-uniform and rare: the record as preprocess wrote it, the reference's expression per pixel
                const uint32_t gid = p.point_list[r0 + e];
                const float4 g0 = p.rec[(size_t)gid * 4], g1 = p.rec[(size_t)gid * 4 + 1];
#pragma unroll
                for (int q = 0; q < PPL; q++) p2q[q] = ref_power_E(g0.x - pxf[q], g0.y - pyf[q], g1.x, g1.y, g1.z, q0.z);
            }
#pragma unroll
            for (int q = 0; q < PPL; q++) {
                if (live[q] == 0ull) continue;                        // wave-uniform: quadrant finished
                const float p2 = p2q[q];                              // = -2 * power
                // "alpha < 1/255" as ONE compare of E against log2(255) (common.h).  No exp for quadrants that nobody passes.
                uint64_t m = __builtin_amdgcn_ballot_w64(p2 <= ALPHA_SKIP_E) & live[q];
                if (DEPTH) {
                    const uint64_t running = __builtin_amdgcn_ballot_w64(e >= resume[q]) & live[q];
                    if (__builtin_amdgcn_inverse_ballot_w64(running)) cnt[q]++;
                    m &= running;
                }
                if (m == 0ull) continue;                              // wave-uniform: nobody sees this Gaussian
                IBGS_LANES_ADD(2, 64); IBGS_LANES_ADD(3, __builtin_popcountll(m));
#ifdef IBGS_COUNT_LANES
                lc_any_ = true;
#endif
                const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(-p2));
                const float aeff = __builtin_amdgcn_inverse_ballot_w64(m) ? alpha : 0.f;
                float aT = aeff * T[q];
                float test_T = T[q] - aT;                             // = T (1 - alpha) up to one rounding, one instruction less
                const uint64_t fin = __builtin_amdgcn_ballot_w64(test_T < 0.0001f) & m;
                if (fin != 0ull) {                                    // rare: some pixels terminate here (not blended, Q7)
                    const bool pf = __builtin_amdgcn_inverse_ballot_w64(fin);
                    aT = pf ? 0.f : aT;
                    test_T = pf ? T[q] : test_T;
                    live[q] &= ~fin;
                    m &= ~fin;
                }
                const bool acc = __builtin_amdgcn_inverse_ballot_w64(m);
                const uint32_t contributor = DEPTH ? cnt[q] : e1v;
                if (!DEPTH) { C[q][0] += q2.x * aT; C[q][1] += q2.y * aT; C[q][2] += q2.z * aT; }
                if (GEO) { Nacc[q][0] += q3.x * aT; Nacc[q][1] += q3.y * aT; Nacc[q][2] += q3.z * aT; }
                // Geo: a pixel feeds its median buffer only until T has dropped to 0.5 AND the "below" half is full -- a handful of
                // contributors; `recm` (an SGPR lane mask like `live`) holds the pixels of the quadrant that still record, and once
                // none of the lanes that blend this Gaussian does, the whole buffer block is skipped (most of a tile's list).
                if ((GEO && (m & recm[q]) != 0ull) || DEPTH) {
                    // ray/plane depth = -dist / denom (forward.cu:439-442).  The geo pass only needs its SIGN inside the
                    // loop (the "depth > 0" gate); the value is recomputed in the epilogue for the <= L buffered entries,
                    // so neither the division nor a depth ring lives in the hot loop.
                    const float denom = q3.x * rayx[q] + q3.y * rayy[q] + q3.z + eps;
                    float dep = 0.f;
                    if (DEPTH) dep = -q1.w / denom;
                    const bool pos = DEPTH ? (dep > 0.0f) : (q1.w * denom < 0.0f);       // sign of -dist / denom (|dist * denom| cannot underflow: |denom| >= 1e-8)
                    const bool hit = acc && pos;
                    const bool front = T[q] > 0.5f;
                    int slot = -1;
                    if (hit && front) slot = before_ptr[q];
                    else if (hit && below_count[q] < below_cap) slot = before_cap + below_count[q];
                    if (DEPTH && slot >= 0 && front) {
                        float oldw = 0.f, oldd = 0.f;
#pragma unroll
                        for (int s = 0; s < (RING_LDS ? 1 : MAXL); s++) if (s == slot) { oldw = bw[q][s]; oldd = bd[q][s]; }
                        tot_w[q] -= oldw; wd_sum[q] -= oldw * oldd;
                    }
                    if (slot >= 0) {
                        if (RING_LDS) { s_bw[q][slot][lane] = aT; s_bc[q][slot][lane] = contributor; }
                        else {
#pragma unroll
                            for (int s = 0; s < (RING_LDS ? 1 : MAXL); s++) if (s == slot) { if (DEPTH) bd[q][s] = dep; bw[q][s] = aT; bc[q][s] = contributor; }
                        }
                        if (front) { const int nb = before_ptr[q] + 1; before_ptr[q] = (nb == before_cap) ? 0 : nb; }   // (ptr + 1) % cap without a division
                        else below_count[q]++;
                        if (DEPTH) { tot_w[q] += aT; wd_sum[q] += aT * dep; }
                    }
                    if (DEPTH && hit && below_count[q] == below_cap) {
                        // forward.cu:484-488: 'break' leaves the current 256-entry round only
                        resume[q] = (e / 256 + 1) * 256;
                    }
                    if (DEPTH && below_cap > 0) {
                        // Once the below-half is full the weighted sums can no longer change (T <= 0.5 from here on), so
                        // the pixel is finished.  The reference keeps blending until T < 1e-4 for nothing; only the
                        // internal final_T / n_contrib of a depth-only pass differ, its output does not.
                        live[q] &= ~__builtin_amdgcn_ballot_w64(hit && below_count[q] == below_cap);
                    }
                    // T only falls and the below-half only fills: a pixel that has both never records again
                    if (GEO) recm[q] &= ~__builtin_amdgcn_ballot_w64(test_T <= 0.5f && below_count[q] >= below_cap);
                }
                T[q] = test_T;
                lastc[q] = acc ? contributor : lastc[q];
            }
#ifdef IBGS_COUNT_LANES
            if (lc_any_) IBGS_LANES_ADD(1, 1);
#endif
        }
        __syncthreads();
        uint64_t anylive = 0ull;
#pragma unroll
        for (int q = 0; q < PPL; q++) anylive |= live[q];
        if (anylive == 0ull) break;
    }

    // ---------------------------------------------------------------- epilogue
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        if (!inside[q]) continue;
        const size_t pix = vbase + (size_t)py[q] * W + px[q];
        p.final_T[pix] = T[q];
        p.n_contrib[pix] = lastc[q];
        if (!DEPTH) {
            p.out_color[pix] = C[q][0] + T[q] * p.cam.bg[0];
            p.out_color[HW + pix] = C[q][1] + T[q] * p.cam.bg[1];
            p.out_color[2 * HW + pix] = C[q][2] + T[q] * p.cam.bg[2];
        }
        if (DEPTH) p.out_depth[pix] = wd_sum[q] / (tot_w[q] + eps);
        if (GEO) { p.out_normal[pix] = Nacc[q][0]; p.out_normal[HW + pix] = Nacc[q][1]; p.out_normal[2 * HW + pix] = Nacc[q][2]; }
    }
    // how far this wave walked the tile's list = what the backward will have to do here (it orders its launch by it, render_bwd.hip): one word
    // per (tile, wave of the tile); meta[10] says how many waves a tile has in this variant
    {
        uint32_t m = 0;
#pragma unroll
        for (int q = 0; q < PPL; q++) m = max(m, inside[q] ? lastc[q] : 0u);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, WAVE));
        if (p.hybrid) {
            // hybrid colour kernel: four words per tile whichever shape walked it (a tile wave leaves the other three at zero)
            if (IPT == 1) { if (lane < 4) p.walked[(size_t)tile * 4 + lane] = lane == 0 ? m : 0u; }
            else if (lane == 0) p.walked[(size_t)tile * 4 + sub] = m;
            if (lane == 0 && tile == 0 && sub == 0) p.meta[10] = 4u;
        }
        else if (lane == 0) { p.walked[(size_t)tile * IPT + sub] = m; if (tile == 0 && sub == 0) p.meta[10] = (uint32_t)IPT; }          // (workgroup 0 may hold no tile under a launch order hint)
        // four flag words per tile whatever the variant (wave `sub` of IPT writes words sub, sub + IPT, ...): non-zero = this wave staged a near-singular conic, i.e. the
        // backward may meet one here (it walks no further than the forward did)
        if (!DEPTH && p.risky && lane < PPL) p.risky[(size_t)tile * 4 + sub + lane * IPT] = riskany != 0ull ? 1u : 0u;
    }
    // The geo epilogue proper runs quadrant after quadrant in a ROLLED loop on values recomputed from (q, lane): by now the blend loop's
    // per-quadrant registers (T, colour, normal sums, ...) are dead, and what stays live is one quadrant's worth of epilogue state --
    // the kernel's register count is the blend loop's.
    if (GEO) {
        // The epilogue takes DECISIONS on what it computes (is the projected point inside the source image, is the source's depth within the
        // threshold): no fused multiply-adds here, so that those coordinates are the oracle's bit for bit whenever the buffered weights are
#pragma clang fp contract(off)
#pragma unroll 1
        for (int q = 0; q < PPL; q++) {
            const int qq = quad0 + q;
            const int pxq = tx0 + (qq & 1) * 8 + (lane & 7), pyq = ty0 + (qq >> 1) * 8 + (lane >> 3);
            if (!(pxq < W && pyq < H)) continue;
            const size_t pix = vbase + (size_t)pyq * W + pxq;
            const float pxfq = (float)pxq, pyfq = (float)pyq;
            const float rayxq = (pxfq - cx) / fx, rayyq = (pyfq - cy) / fy;          // as before the blend loop: the same two divisions
            const float inv_fx = 1.0f / fx, inv_fy = 1.0f / fy;
            const float pdx = pxfq - cx, pdy = pyfq - cy;
            // (1) the buffered contributors: weight and ray/plane depth per slot, their weighted mean = the median depth.  The depth takes
            // the contributor number's place in the LDS column (a register-held ring keeps both in registers): step (2) walks the slots
            // in a rolled loop and the epilogue needs no more registers than the blend loop
            float dv[RING_LDS ? 1 : MAXL];
            float tw = 0.f, med = 0.f;
            uint32_t lo = RING_LDS ? s_bc[q][0][lane] : bc[q][0], hi = lo;     // Q4: slot 0 even when empty
#pragma unroll 1
            for (int s = 0; s < L; s++) {
                float w = 0.f; uint32_t c = 0;
                if (RING_LDS) { w = s_bw[q][s][lane]; c = s_bc[q][s][lane]; }
                else {
#pragma unroll
                    for (int k = 0; k < (RING_LDS ? 1 : MAXL); k++) if (k == s) { w = bw[q][k]; c = bc[q][k]; }
                }
                p.slot_c[(size_t)s * HW + pix] = (w == 0.0f) ? 0u : c;      // the backward's window pass starts from these (render_bwd.hip)
                float d = 0.f;
                if (w != 0.0f) {
                    // depth of buffered contributor c (1-based list position): same expression as the blend loop would use
                    const uint32_t gid = p.point_list[r0 + c - 1u];
                    const float4 g1 = p.rec[(size_t)gid * 4 + 1], g3 = p.rec[(size_t)gid * 4 + 3];
                    d = -g1.w / (g3.x * rayxq + g3.y * rayyq + g3.z + eps);
                    tw += w; med += w * d;
                    lo = min(lo, c); hi = max(hi, c);
                }
                if (RING_LDS) s_bc[q][s][lane] = __float_as_uint(d);
                else {
#pragma unroll
                    for (int k = 0; k < (RING_LDS ? 1 : MAXL); k++) if (k == s) dv[k] = d;
                }
            }
            p.low_high[2 * pix] = lo; p.low_high[2 * pix + 1] = hi; p.sum_w[pix] = tw;
            med /= (tw + eps);
            const float mX = pdx * med * inv_fx, mY = pdy * med * inv_fy, mZ = med;
            const float* vm = p.cam.vm;
            const float qx = mX - vm[12], qy = mY - vm[13], qz = mZ - vm[14];
            const float wx = vm[0] * qx + vm[1] * qy + vm[2] * qz;
            const float wy = vm[4] * qx + vm[5] * qy + vm[6] * qz;
            const float wz = vm[8] * qx + vm[9] * qy + vm[10] * qz;
            float rd0 = wx - p.cam.campos[0], rd1 = wy - p.cam.campos[1], rd2 = wz - p.cam.campos[2];
            const float rl = sqrtf(rd0 * rd0 + rd1 * rd1 + rd2 * rd2) + eps;
            rd0 /= rl; rd1 /= rl; rd2 /= rl;
            p.out_camera_ray[pix] = rd0; p.out_camera_ray[HW + pix] = rd1; p.out_camera_ray[2 * HW + pix] = rd2;

            // (2) source by source: is the median point seen by the source at a consistent depth (forward.cu:596-663)?  Only then are
            // the buffered points warped into it -- the reference warps into every source first and keeps the valid ones; the sums
            // are formed in the same order, the outputs are identical, and four accumulators are live instead of twenty.
            int nvalid = 0, first_ok = 0; float min_err = 1.0f;
#pragma unroll 1
            for (int si = 0; si < p.n_src; si++) {
                if (nvalid >= IBGS_MAX_SRC) break;
                const float* r = p.ref_to_src + 16 * si;
                const float r0_ = r[0], r1_ = r[1], r2_ = r[2], r3_ = r[3], r4_ = r[4], r5_ = r[5], r6_ = r[6], r7_ = r[7], r8_ = r[8], r9_ = r[9], r10_ = r[10], r11_ = r[11];          // wave-uniform: scalar registers
                float wdep = 0.0f, err, tzm, izm;
                {
                    const float tx = r0_ * mX + r1_ * mY + r2_ * mZ + r3_ * 1.0f;
                    const float ty = r4_ * mX + r5_ * mY + r6_ * mZ + r7_ * 1.0f;
                    tzm = r8_ * mX + r9_ * mY + r10_ * mZ + r11_ * 1.0f;
                    izm = 1.0f / (tzm + eps);
                    const float u = tx * fx * izm + cx, v = ty * fy * izm + cy;
                    if (u >= 0.0f && u <= (float)(W - 1) && v >= 0.0f && v <= (float)(H - 1))
                        wdep = tex_depth(p.src_depths + (size_t)(si == 0 ? p.ds0 : si == 1 ? p.ds1 : si == 2 ? p.ds2 : si == 3 ? p.ds3 : p.ds4) * HW, W, H, u + 0.5f, v + 0.5f, p.tex_quant);
                    err = fabsf(wdep - tzm) * izm;
                }
                if (!(wdep > 0.0f && err < p.thr)) continue;
                float c0 = 0.f, c1 = 0.f, c2 = 0.f, tws = 0.f;
                const float4* img = p.src_rgba + (size_t)si * HW;
#pragma unroll 1
                for (int s = 0; s < L; s++) {
                    float w = 0.f, d = 0.f;
                    if (RING_LDS) { w = s_bw[q][s][lane]; d = __uint_as_float(s_bc[q][s][lane]); }
                    else {
#pragma unroll
                        for (int k = 0; k < (RING_LDS ? 1 : MAXL); k++) if (k == s) { w = bw[q][k]; d = dv[k]; }
                    }
                    const float X = pdx * d * inv_fx, Y = pdy * d * inv_fy, Z = d;
                    const float tx = r0_ * X + r1_ * Y + r2_ * Z + r3_ * 1.0f;
                    const float ty = r4_ * X + r5_ * Y + r6_ * Z + r7_ * 1.0f;
                    const float tz = r8_ * X + r9_ * Y + r10_ * Z + r11_ * 1.0f;
                    const float iz = 1.0f / (tz + eps);
                    const float u = tx * fx * iz + cx, v = ty * fy * iz + cy;
                    if (w != 0.0f && u >= 0.0f && u <= (float)(W - 1) && v >= 0.0f && v <= (float)(H - 1)) {
                        const float4 col = tex_rgba(img, W, H, u + 0.5f, v + 0.5f, p.tex_quant);
                        c0 += w * col.x; c1 += w * col.y; c2 += w * col.z; tws += w;
                    }
                }
                const float iw = 1.0f / (tws + eps);
                const float* sp = p.src_cam_pos + 3 * si;
                p.out_cam_feat[((size_t)nvalid * 4 + 0) * HW + pix] = p.cam.campos[0] - sp[0];
                p.out_cam_feat[((size_t)nvalid * 4 + 1) * HW + pix] = p.cam.campos[1] - sp[1];
                p.out_cam_feat[((size_t)nvalid * 4 + 2) * HW + pix] = p.cam.campos[2] - sp[2];
                p.out_warped[((size_t)nvalid * 3 + 0) * HW + pix] = c0 * iw;
                p.out_warped[((size_t)nvalid * 3 + 1) * HW + pix] = c1 * iw;
                p.out_warped[((size_t)nvalid * 3 + 2) * HW + pix] = c2 * iw;
                float s0 = wx - sp[0], s1 = wy - sp[1], s2 = wz - sp[2];
                const float sl = sqrtf(s0 * s0 + s1 * s1 + s2 * s2) + eps;
                s0 /= sl; s1 /= sl; s2 /= sl;
                p.out_cam_feat[((size_t)nvalid * 4 + 3) * HW + pix] = s0 * rd0 + s1 * rd1 + s2 * rd2;
                if (si == 0) first_ok = 1;
                p.valid_idx[(size_t)nvalid * HW + pix] = si;
                p.valid_w[(size_t)nvalid * HW + pix] = tws;
                nvalid++;
                min_err = fminf(min_err, err);
            }
            if (nvalid <= IBGS_MAX_SRC - 1) p.valid_idx[(size_t)nvalid * HW + pix] = -1;
            // unused source slots and the mask are written too, so the caller needs no 36-plane memset per frame
            // (the reference zero-fills every output plane on each call, rasterize_points.cu:80-90)
#ifndef IBGS_DIAG_NO_ZERO_FILL          // (diagnostic build, profiles/r06_geo_fwd_split.txt: what the zero planes cost; its outputs are NOT valid)
            for (int k = nvalid; k < IBGS_MAX_SRC; k++) {
#pragma unroll
                for (int ch = 0; ch < 4; ch++) p.out_cam_feat[((size_t)k * 4 + ch) * HW + pix] = 0.f;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) p.out_warped[((size_t)k * 3 + ch) * HW + pix] = 0.f;
            }
#endif
            p.out_mask[pix] = first_ok;
            p.out_min_depth_diff[pix] = min_err;
            p.out_depth[pix] = med;
        }
    }
    IBGS_TRACE_END(g_trace_fwd);
    IBGS_LANES_FLUSH(g_lanes_fwd);
}

template <int MODE, int PPL, int MAXL>
__global__ void __launch_bounds__(64, (MODE == 2 && PPL == 4 && MAXL == 4) ? 5 : 1) render_fwd_kernel(FwdParams p)
{
    __shared__ float4 s_rec[(MODE == MODE_GEO) ? 4 : 3][(MODE == MODE_GEO) ? 16 : WAVE];
    constexpr int IPT = 4 / PPL;
    int tile, sub;
    if (((MODE == MODE_COLOR && PPL == 4) || (MODE == MODE_GEO && PPL == 2)) && p.order) {
        // launched over the slots of a tile order (ibgs_forward_args::tile_order_hint), IPT consecutive workgroups per slot: the hinted tile when the
        // hint was found valid, else tile = slot
        uint32_t t = blockIdx.x / IPT;
        if (p.meta[11] == 1u) { t = p.order[t]; if (t != 0xFFFFFFFFu) t &= ~ORDER_SPLIT_BIT; }
        if (t >= (uint32_t)p.ntiles) return;          // (0xFFFFFFFF: an empty slot)
        tile = (int)t; sub = (int)(blockIdx.x % IPT);
    } else if (!tile_map_item(p.tmap, blockIdx.x, p.cam.gx, p.ntiles / p.cam.gx, IPT, tile, sub)) return;
    render_fwd_body<MODE, PPL, MAXL>(p, tile, sub, s_rec);
}

// ---- hybrid colour kernel: frames of fewer tiles than the chip has wave slots --------------------------------------------------------------
// One wave per tile leaves SIMDs idle or with a single wave (which issues at half the rate of two: probe_xlane), one wave per 8 x 8 quadrant fills
// the chip but repeats the per-entry work that the four quadrants of a tile share (1.35 x the instructions).  Which is faster depends on the frame
// AND on the tile (tools/sweep_wave_shape.py: tile waves win from ~1 500 even tiles up, quadrant waves below; one heavy list in a frame of light
// ones is walked four times faster by four waves).  So the choice is made per tile, on the device: four workgroups per tile are launched; where
// the tile's work exceeds HYBRID_THETA per cent of a SIMD's fair share of the frame's (common.h) each of them takes a quadrant, elsewhere the first
// one takes the tile and the other three leave at once.  The tiles' FIRST workgroups are the first of the launch, the other three follow behind
// them all (hybrid_item).  Any choice gives the same image BIT FOR BIT (both bodies blend a pixel's list in the same order with the same
// operations, TILE_Q above): the hint that may steer the choice stays a pure performance matter.
__global__ void __launch_bounds__(64, 8) render_fwd_color_hybrid_kernel(FwdParams p)
{
    __shared__ float4 s_rec[3][WAVE];
    int tile, sub; bool split;
    if (p.order && p.meta[11] == 1u) {
        // the same camera's last backward: its balanced order, and per tile its choice of shape -- made on what counts, how far the list was WALKED (with
        // trained opacities the length of a list says little about that)
        if (!hybrid_item(blockIdx.x, p.hybrid_grid1, p.order, tile, sub, split)) return;
    } else {
        if (p.order) {          // a hint that did not pass the check: slot = tile
            const int idx = ((int)blockIdx.x - p.hybrid_grid1) >> 3;          // (as hybrid_item)
            tile = blockIdx.x < (unsigned)p.hybrid_grid1 ? (int)blockIdx.x : (idx / 3) * 8 + (int)(blockIdx.x & 7u);
            sub = blockIdx.x < (unsigned)p.hybrid_grid1 ? 0 : idx % 3 + 1;
            if (tile >= p.ntiles) return;
        }
        else if (!hybrid_item_mapped(p.tmap, blockIdx.x, p.hybrid_grid1, p.cam.gx, p.ntiles / p.cam.gx, tile, sub)) return;
        // no measurement to go by: four quadrant waves for every tile, what the library did on such frames before the hybrid kernels.  (The length of the
        // tile's list is no substitute: right on even scenes with initial opacities, but with trained opacities most of a long list is never walked and the
        // heavy tiles are the UNsaturated ones -- 800 x 800, trained: 0.28 ms by length against 0.16 ms for all-quadrant and for the backward's flags.)
        split = true;
    }
    if (split) {
        if (sub == 0) render_fwd_body<MODE_COLOR, 1, 4, 0>(p, tile, 0, s_rec);
        else if (sub == 1) render_fwd_body<MODE_COLOR, 1, 4, 1>(p, tile, 1, s_rec);
        else if (sub == 2) render_fwd_body<MODE_COLOR, 1, 4, 2>(p, tile, 2, s_rec);
        else render_fwd_body<MODE_COLOR, 1, 4, 3>(p, tile, 3, s_rec);
    }
    else if (sub == 0) render_fwd_body<MODE_COLOR, 4, 4>(p, tile, 0, s_rec);
}

int launch_render_forward(hipStream_t s, const ibgs_forward_args& a, const GeomState& g, const BinState& b,
                          const ImgState& im, const float4* src_rgba)
{
    FwdParams p;
    p.ranges = im.ranges; p.point_list = b.point_list; p.rec = reinterpret_cast<const float4*>(g.rec);
    p.cam = make_cam(a.viewmatrix, a.projmatrix, a.campos, a.bg, a.tanfovx, a.tanfovy, a.W, a.H);
    p.ntiles = p.cam.gx * p.cam.gy;
    p.n_src = a.n_src; p.L = a.buffer_length; p.thr = a.depth_error_threshold;
    p.tex_quant = (a.flags & IBGS_FLAG_TEX_QUANT) ? 1 : 0;
    p.power_skip = (a.flags & IBGS_FLAG_NO_REF_POWER_SKIP) ? 0 : 1;
    p.ref_to_src = a.ref_to_src; p.src_cam_pos = a.src_cam_pos; p.src_rgba = src_rgba; p.src_depths = a.src_depths;
    {   // the depth plane of every source: its own number, or the caller's table slot
        const bool slots = (a.flags & IBGS_FLAG_SRC_DEPTH_SLOTS) != 0;
        p.ds0 = slots ? a.src_depth_slot[0] : 0; p.ds1 = slots ? a.src_depth_slot[1] : 1; p.ds2 = slots ? a.src_depth_slot[2] : 2;
        p.ds3 = slots ? a.src_depth_slot[3] : 3; p.ds4 = slots ? a.src_depth_slot[4] : 4;
    }
    p.final_T = im.final_T; p.n_contrib = im.n_contrib; p.sum_w = im.sum_w; p.low_high = im.low_high;
    p.valid_idx = im.valid_idx; p.valid_w = im.valid_w; p.slot_c = im.slot_c; p.meta = im.meta; p.walked = im.tile_walked; p.risky = im.tile_risky; p.order = nullptr;
    p.out_color = a.out_color; p.out_normal = a.out_normal; p.out_depth = a.out_depth; p.out_cam_feat = a.out_cam_feat;
    p.out_warped = a.out_warped; p.out_min_depth_diff = a.out_min_depth_diff; p.out_camera_ray = a.out_camera_ray;
    p.out_mask = a.out_mask;
    p.hybrid = 0;
    p.n_views = a.n_views > 1 ? a.n_views : 1; p.gyv = p.cam.gy;
    for (int v = 0; v < IBGS_MAX_VIEWS; v++) {
        p.fxv[v] = (v < p.n_views && p.n_views > 1) ? a.W / (2.0f * a.view_tanfovx[v]) : p.cam.fx;
        p.fyv[v] = (v < p.n_views && p.n_views > 1) ? a.H / (2.0f * a.view_tanfovy[v]) : p.cam.fy;
    }
    p.ntiles *= p.n_views;
    const int nt = p.ntiles;
    const int gx = p.cam.gx, gyt = nt / gx;          // (stacked) tile grid
    // measured (profiles/r03_tile_map.txt): the colour / depth kernels are VALU-bound on every layout and within +-1.5 % of each other;
    // 8 x 8-tile blocks are the fastest on the uniform scene and equal to round-robin on a clustered one.  The geo kernel's epilogue gathers from the packed source images: 8 x 8-tile blocks cut its L2 <-> fabric traffic
    // from 2.39 GB to 1.37 GB (= the algorithmic bytes) and its time by 3 %
    p.tmap = TileMap{TMAP_BLOCK, 1, 8, 8};          // colour and geo alike (docs/EXPERIMENTS.md section 7 keeps the sweep over the other layouts)
    auto grid = [&](int ipt) { return dim3((unsigned)tile_map_grid(p.tmap, gx, gyt, ipt)); };
    if (a.render_depth_only && !a.render_geo) {
        if (a.buffer_length <= 4) hipLaunchKernelGGL((render_fwd_kernel<MODE_DEPTH, 4, 4>), grid(1), dim3(64), 0, s, p);
        else hipLaunchKernelGGL((render_fwd_kernel<MODE_DEPTH, 4, 8>), grid(1), dim3(64), 0, s, p);
    } else if (a.render_geo) {
        // geo: half tiles (two quadrants per lane) on large frames, single quadrants on small ones; IBGS_FLAG_*_WAVES force either
        const bool half = (a.flags & IBGS_FLAG_QUADRANT_WAVES) ? false : ((a.flags & IBGS_FLAG_TILE_WAVES) ? true : nt >= 4096);
        if (half && a.buffer_length <= 4 && a.tile_order_hint) {
            p.order = a.tile_order_hint;
            hipLaunchKernelGGL((render_fwd_kernel<MODE_GEO, 2, 4>), dim3(2u * (unsigned)((p.ntiles + ORDER_CLASSES - 1) / ORDER_CLASSES * ORDER_CLASSES)), dim3(64), 0, s, p);
        }
        else if (half && a.buffer_length <= 4) hipLaunchKernelGGL((render_fwd_kernel<MODE_GEO, 2, 4>), grid(2), dim3(64), 0, s, p);
        else if (a.buffer_length <= 4) hipLaunchKernelGGL((render_fwd_kernel<MODE_GEO, 1, 4>), grid(4), dim3(64), 0, s, p);
        else hipLaunchKernelGGL((render_fwd_kernel<MODE_GEO, 1, 8>), grid(4), dim3(64), 0, s, p);
    } else {
        // Small frames: one wave per 8x8 quadrant instead of per tile, otherwise the chip (1024 SIMDs x 8 waves) stays
        // mostly empty and every wave walks its list alone (800x800 has 2500 tiles).
        const bool small = (a.flags & IBGS_FLAG_QUADRANT_WAVES) ? true : ((a.flags & IBGS_FLAG_TILE_WAVES) ? false : nt < hybrid_max_tiles());
        const bool forced = (a.flags & (IBGS_FLAG_QUADRANT_WAVES | IBGS_FLAG_TILE_WAVES)) != 0;
        if (small && !forced && p.n_views <= 1 && nt >= HYBRID_MIN_TILES) {
            p.hybrid = 1;
            if (a.tile_order_hint) {          // (the same camera's last backward order, as for the tile-wave kernel below)
                p.order = a.tile_order_hint;
                p.hybrid_grid1 = (p.ntiles + ORDER_CLASSES - 1) / ORDER_CLASSES * ORDER_CLASSES;
                hipLaunchKernelGGL(render_fwd_color_hybrid_kernel, dim3(4u * (unsigned)p.hybrid_grid1), dim3(64), 0, s, p);
            } else {
                p.hybrid_grid1 = tile_map_grid(p.tmap, gx, gyt, 1);
                hipLaunchKernelGGL(render_fwd_color_hybrid_kernel, dim3((unsigned)(p.hybrid_grid1 + tile_map_grid(p.tmap, gx, gyt, 3))), dim3(64), 0, s, p);
            }
        }
        else if (small) hipLaunchKernelGGL((render_fwd_kernel<MODE_COLOR, 1, 4>), grid(4), dim3(64), 0, s, p);
        else if (a.tile_order_hint && p.n_views <= 1) {
            p.order = a.tile_order_hint;
            hipLaunchKernelGGL((render_fwd_kernel<MODE_COLOR, 4, 4>), dim3((unsigned)((p.ntiles + ORDER_CLASSES - 1) / ORDER_CLASSES * ORDER_CLASSES)), dim3(64), 0, s, p);
        }
        else hipLaunchKernelGGL((render_fwd_kernel<MODE_COLOR, 4, 4>), grid(1), dim3(64), 0, s, p);
    }
    IBGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace ibgs

#ifdef IBGS_COUNT_LANES
extern "C" int ibgs_debug_lanes_fwd(unsigned long long* dst, int reset)
{
    int rc = (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(ibgs::g_lanes_fwd), sizeof(unsigned long long) * 4, 0, hipMemcpyDeviceToHost);
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(ibgs::g_lanes_fwd), z, sizeof(z), 0, hipMemcpyHostToDevice); }
    return rc;
}
#endif
#ifdef IBGS_TRACE_WAVES
extern "C" int ibgs_debug_trace_fwd(void* dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(ibgs::g_trace_fwd), bytes < sizeof(uint4) * ibgs::IBGS_TRACE_MAX ? bytes : sizeof(uint4) * ibgs::IBGS_TRACE_MAX, 0, hipMemcpyDeviceToHost);
}
#endif
