/*
 * ibgs_rast.h -- C ABI of libibgs_rast.so, the MI355X (gfx950) plane rasterizer for IBGS.
 *
 * This is the drop-in boundary for the reference's native rasterizer library
 * (/root/reference/submodules/diff-plane-rasterization/, "DPR/" below):
 *
 *   ibgs_forward       replaces CudaRasterizer::Rasterizer::forward   (DPR/cuda_rasterizer/rasterizer.h:98-144,
 *                      rasterizer_impl.cu:320-515) as called by RasterizeGaussiansCUDA (DPR/rasterize_points.cu:37-160)
 *   ibgs_backward      replaces CudaRasterizer::Rasterizer::backward  (rasterizer.h:146-203, rasterizer_impl.cu:519-666)
 *                      as called by RasterizeGaussiansBackwardCUDA (rasterize_points.cu:162-271)
 *   ibgs_mark_visible  replaces CudaRasterizer::Rasterizer::markVisible (rasterizer.h:91-96, rasterizer_impl.cu:258-270)
 *   ibgs_required_*    replace  CudaRasterizer::required<GeometryState|ImageState|BinningState>
 *                      (rasterizer_impl.h:69-76) -- arena sizes for caller-owned scratch
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous memory unless its name starts with "host_";
 *     fp32 / int32 element types as named; a NULL input pointer means "not provided"
 *     (the reference's empty-tensor convention, DPR/diff_plane_rasterization/__init__.py:304-316);
 *   - `stream` is a hipStream_t passed as void*; every kernel and copy is issued on it;
 *   - scratch arenas (geom / binning / img) are owned by the caller and must stay alive and
 *     unmodified between a forward and its backward (the reference keeps them in the autograd
 *     ctx, DPR/diff_plane_rasterization/__init__.py:133-140); the library keeps no state;
 *   - ibgs_backward overwrites EVERY element of every gradient output it is given (zeros for Gaussians
 *     with radius 0), so they need not be cleared first (the reference zero-fills them,
 *     rasterize_points.cu:209-219); only the grad_acc scratch must arrive zeroed;
 *   - return value: >= 0 on success (ibgs_forward: the number of rendered (Gaussian, tile)
 *     instances R), < 0 = -(IBGS_ERR_*). ibgs_last_error() returns a static message.
 */
#ifndef IBGS_RAST_H
#define IBGS_RAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IBGS_MAX_SRC 5           /* DPR/cuda_rasterizer/auxiliary.h:22-23 (MAX_M, M) */
#define IBGS_MAX_BUFFER_LENGTH 8 /* auxiliary.h:21 */
#define IBGS_MAX_VIEWS 8          /* batched depth-only passes (ibgs_forward_args.n_views) */
#define IBGS_TILE 16             /* config.h: BLOCK_X, BLOCK_Y */

#define IBGS_ERR_INVALID 1  /* bad argument combination */
#define IBGS_ERR_HIP 2      /* HIP runtime error (message has the HIP string) */
#define IBGS_ERR_ALLOC 3    /* arena callback returned NULL / arena too small */

/* Arena callback: must return a device pointer to at least `bytes` bytes (128-B aligned is
 * enough).  Mirrors std::function<char*(size_t)> of the reference (rasterizer.h:101-103,
 * rasterize_points.cu:29-35). */
typedef char* (*ibgs_alloc_fn)(size_t bytes, void* user);

/* Option flags (ibgs_forward_args.flags / ibgs_backward_args.flags) */
#define IBGS_FLAG_DEBUG 1u      /* synchronise + check after every stage (auxiliary.h:170-177) */
#define IBGS_FLAG_TEX_QUANT 2u  /* emulate the CUDA texture unit's 8-bit filter weights */
#define IBGS_FLAG_CLEAR_GRAD_ACC 8u /* ibgs_backward re-zeroes every grad_acc row it consumed, so a caller may keep ONE
                                      zeroed scratch alive across steps instead of clearing P x 64 B per call */
#define IBGS_FLAG_NO_TILE_CULL 4u /* emit the reference's full AABB tile lists (rasterizer_impl.cu:205-225) instead of
                                     dropping tiles that provably fail the alpha >= 1/255 test; outputs are identical */

#define IBGS_FLAG_TILE_WAVES 32u     /* blend kernels: always the coarse decomposition -- one wave per 16x16 tile (colour) or per half
                                        tile (geo); default: only when the frame has >= 4096 tiles */
#define IBGS_FLAG_QUADRANT_WAVES 64u /* blend kernels: always one wave per 8x8 quadrant.  Default for frames of fewer than 4096 tiles: geo passes
                                        this; colour passes choose PER TILE on the device -- one wave, or four quadrant waves for a tile whose
                                        work exceeds half of a SIMD's fair share of the frame's (the backward goes by how far the forward walked the
                                        tile's list; the forward by that backward's choice when it is given tile_order_hint, else four waves everywhere) */

#define IBGS_PLANE_NONE 0
#define IBGS_PLANE_LEARNT 1
#define IBGS_PLANE_SMALLEST_AXIS 2

#define IBGS_FLAG_DETERMINISTIC 128u /* ibgs_backward only: no float atomics.  The render backward stores its per-(Gaussian, tile) sums in a
                                       slab indexed by the entry's position in the sorted list; the (Gaussian id, position) pairs are
                                       radix-sorted by id (stable: positions stay ascending) and every Gaussian's rows are summed
                                       sequentially in that order (SURVEY 7.2 "alternative without atomics"; replaces the atomicAdds of
                                       backward.cu:673, 770, 793-804).  Gradients are then bit-identical from run to run; they differ from
                                       the default mode only in summation order.  Needs det_scratch.  For CI, ~0.7 ms slower at C3. */
#define IBGS_FLAG_TEX_PACKED 256u /* `tex` already holds the packed RGBA of exactly these src_images -- ibgs_backward: the ibgs_forward of this call pair wrote it
                                     and nothing touched the buffer since; ibgs_forward: an earlier call packed the same, unmodified image stack
                                     (a trainer's source images do not change between steps) -- skip the pack (T1 once per source set) */
#define IBGS_FLAG_SH_FACTORED 16u /* ibgs_backward only, view-parallel training: dL/dsh of ONE view is the outer product
                                     basis(dir) x dL/dRGB (backward.cu:114-160), so leave dL_dsh unwritten (may be NULL) and
                                     write the clamp-masked dL/dRGB (P x 3) to dL_dcolors; after the ranks exchanged those
                                     3 floats instead of 3 M, ibgs_sh_grad_from_views rebuilds the summed dL/dsh */

#define IBGS_FLAG_NO_REF_POWER_SKIP 512u /* both passes: do NOT reproduce the reference's `if (power > 0.0f) continue;` (forward.cu:420,
                                           backward.cu:645).  By default the blend kernels evaluate the reference's expression per pixel,
                                           uncontracted, for every Gaussian whose conic is within 1e-3 of singular (b^2 > 0.999 a c; csrc/common.h:
                                           conic_takes_ref_power / BLEND_REF_POWER_RISK) and drop the pairs it drops -- the test itself can only fire
                                           within 1e-5 (conic_is_risky / POWER_RISK: those are also exempt from the tile cull); the two decades around
                                           them take the branch so that their decisions round as the reference's do.  With this flag every Gaussian
                                           takes the fast path and such pairs are blended with alpha ~= opacity */

#define IBGS_FLAG_REF_ARITH 4096u /* ibgs_backward only (SURVEY Q1 as a switch).  The pairs of a Gaussian whose conic is within 1e-3 of singular (csrc/common.h:
                                     conic_takes_ref_power) are always evaluated with G = exp(power) at libm accuracy (backward.cu:648) and T / (1 - alpha) as an IEEE division
                                     (:654).  By default their sums are formed in the well-conditioned form dL/dcov2D = 0.5 sum q l l^T, l = conic d -- algebraically the reference's
                                     chain (backward.cu:405-420), without its cancellation: 3-10 x closer to a float64 evaluation than ANY fp32 evaluation of the reference's own
                                     expressions (profiles/r06_ref_arith_ab.txt).  With this flag they are formed exactly as the reference forms them: the eight per-pair quantities of
                                     backward.cu:779-804 in its association, uncontracted, and the chain of :405-420 on their sums.  Every other Gaussian: unchanged */
#define IBGS_FLAG_SRC_DEPTH_SLOTS 8192u /* ibgs_forward only: src_depths is a table of planes, source m reads plane src_depth_slot[m] (ibgs_forward_args) */
#define IBGS_FLAG_NO_ABS_GRAD 1024u /* ibgs_backward only: dL_dmean2D_abs is not wanted (it may be NULL and is not written).  It is the densification statistic of
                                       train.py:400-410 (sum over pixels of |dL/dmean2D| per Gaussian, backward.cu:793-804): nobody reads it after
                                       densify_until_iter or at test time.  The colour blend then skips the two |.| moments (conic x d per quadrant, two fma per
                                       pair); every other gradient is unchanged bit for bit up to the order of the float atomics */

typedef struct ibgs_forward_args {
    void* stream;
    /* problem size */
    int32_t P;         /* number of Gaussians */
    int32_t D;         /* active SH degree (0..3) */
    int32_t M;         /* SH coefficients per Gaussian in memory (shs is P x M x 3) */
    int32_t W, H;
    /* per-Gaussian inputs */
    const float* means3D;        /* P x 3 */
    const float* shs;            /* P x M x 3 or NULL */
    const float* colors_precomp; /* P x 3 or NULL */
    const float* opacities;      /* P */
    const float* scales;         /* P x 3 or NULL */
    const float* rotations;      /* P x 4 (w,x,y,z) or NULL */
    const float* cov3D_precomp;  /* P x 6 or NULL */
    const float* all_map;        /* P x 5 or NULL */
    float scale_modifier;
    /* camera */
    const float* bg;         /* 3 */
    const float* viewmatrix; /* 16, transposed world->view (scene/cameras.py:102) */
    const float* projmatrix; /* 16, transposed full projection (cameras.py:104) */
    const float* campos;     /* 3 */
    float tanfovx, tanfovy;
    /* source views (geo path) */
    int32_t n_src;              /* 1..IBGS_MAX_SRC */
    const float* ref_to_src;    /* n_src x 16, row-major true matrices */
    const float* src_cam_pos;   /* n_src x 3 */
    const float* src_images;    /* n_src x 3 x H x W */
    const float* src_depths;    /* n_src x 1 x H x W */
    int32_t buffer_length;      /* 1..IBGS_MAX_BUFFER_LENGTH */
    float depth_error_threshold;
    /* modes */
    int32_t prefiltered;        /* signature compatibility only.  The reference traps the kernel when a Gaussian is near-culled although the
                                   caller set this (auxiliary.h:158-166; its callers always pass false); here such a Gaussian is just culled */
    int32_t render_geo;
    int32_t render_depth_only;
    uint32_t flags;
    /* scratch arenas */
    char* geom;  size_t geom_bytes;    /* >= ibgs_required_geom(P) */
    char* img;   size_t img_bytes;     /* >= ibgs_required_img(W, H) */
    ibgs_alloc_fn binning_alloc;       /* called with ibgs_required_binning(R, W, H) (or of rendered_hint, see below) */
    void* binning_user;
    char* tex;   size_t tex_bytes;     /* render_geo only: >= ibgs_required_tex(n_src, W, H); contents are
                                          transient (packed RGBA source images), may be shared between calls */
    /* outputs: every element of the planes a mode produces is written (unused source slots of out_cam_feat /
     * out_warped as zeros), nothing needs to arrive zeroed.  Pointers the mode does not write may be NULL. */
    float* out_color;          /* 3 x H x W */
    int32_t* radii;            /* P */
    float* out_normal;         /* 3 x H x W   (render_geo) */
    float* out_depth;          /* 1 x H x W   (render_geo or render_depth_only) */
    float* out_cam_feat;       /* 20 x H x W  (render_geo) */
    float* out_warped;         /* 15 x H x W  (render_geo) */
    float* out_min_depth_diff; /* 1 x H x W   (render_geo) */
    float* out_camera_ray;     /* 3 x H x W   (render_geo) */
    int32_t* out_mask;         /* 1 x H x W   (render_geo) */
    /* Optional upper-bound guess for the return value R (e.g. 1.25 x the R of the previous call with this
     * camera/scene), 0 = none.  R is produced on the device; the reference (rasterizer_impl.cu:404-410) and this
     * library without a hint stop the host until it has been copied back, because the binning arena is sized from
     * it.  With a hint, binning_alloc is called for the hint right away, binning and rendering are enqueued sized
     * for it (they read the true count from device memory), and only then does the call wait for R, so the GPU does
     * not idle during the round trip.  Results are identical.  If the true R exceeds the hint the call drains the
     * stream, calls binning_alloc again with the exact size and repeats binning + rendering.  Ignored when
     * IBGS_FLAG_DEBUG is set.  ibgs_backward needs nothing extra: it locates the sorted list independently of
     * the size the arena was carved for. */
    int64_t rendered_hint;
    /* SURVEY 8(f) row 1 -- plane-map glue fused into preprocess.  plane_mode != 0 replaces all_map (which must then
     * be NULL): the library derives [n_cam, 1, |d_cam|] per Gaussian itself instead of the ~10 torch kernels of
     * gaussian_renderer/__init__.py:304-316 + scene/gaussian_model.py:156-173.
     *   IBGS_PLANE_LEARNT:        plane_normal = raw `_normal` (P x 3, not normalised), plane_offset = raw `_offset` (P) or NULL
     *   IBGS_PLANE_SMALLEST_AXIS: normal = column argmin(scales) of R(rotations); needs scales + rotations */
    const float* plane_normal;
    const float* plane_offset;
    int32_t plane_mode;
    /* SURVEY 8(f) row 2 -- batched depth-only passes.  n_views >= 2 (<= IBGS_MAX_VIEWS) renders the median depth of
     * n_views cameras of equal W x H in ONE pass: the views are laid out as one tall tile grid, so the depth sort, the
     * binning and the blend each run once over n_views x P instances instead of n_views times (the reference loops
     * render_depth over the source views, gaussian_renderer/__init__.py:245-253).  Requires render_depth_only and
     * plane_mode != 0.  viewmatrix / projmatrix / campos hold n_views stacked blocks (n x 16, n x 16, n x 3);
     * view_tanfovx/y the per-view values (tanfovx / tanfovy are ignored); out_depth is n_views x H x W, radii
     * n_views x P; geom must hold ibgs_required_geom(n_views * P), img ibgs_required_img(W, n_views * 16 * ceil(H/16)).
     * Each view's result is bit-identical to its single-view pass.  No backward. */
    int32_t n_views;
    float view_tanfovx[8];
    float view_tanfovy[8];
    /* Launch order hint for the colour blend kernel (optional, may be NULL; device pointer to ibgs_tile_order_slots(W, H) words).
     * ibgs_backward of a colour pass leaves the order in which it launched its tiles -- balanced over the SIMDs by how far the forward
     * walked every tile's list -- at img + ibgs_img_offset(W, H, "tile_order") and in ibgs_backward_args.tile_order_out.  Handed to a later ibgs_forward of the SAME
     * camera (whose lists saturate where they did before), lets the forward launch balanced too; the forward cannot know its own work
     * in advance.  Bit 31 of a tile's word: that backward found the tile heavy enough for four quadrant waves (frames of fewer than 4096 tiles).
     * Performance only: the words are checked on the device (every tile exactly once, 0xFFFFFFFF = empty slot) and anything
     * else -- a stale order, garbage -- is ignored; the buffer itself must hold ibgs_tile_order_slots(W, H) words (all of them are read).  Used by
     * the variants with one wave per tile / half tile (large frames: colour and render_geo passes) and by the per-tile choice of small colour frames,
     * not by render_depth_only. */
    const uint32_t* tile_order_hint;
    /* Optional (may be NULL / 0): a binning arena the caller allocated beforehand for rendered_hint, >= ibgs_required_binning(rendered_hint, W, H) bytes.
     * The hinted pass then uses it instead of calling binning_alloc (a call back into the host language: ~10 us through ctypes); a forward without a
     * hint, and the repeated pass after a too small hint, still call binning_alloc.  The arena the lists ended up in is `binning` iff the return value
     * R <= rendered_hint and this field was large enough. */
    char* binning; size_t binning_bytes;
    /* Optional: the SH coefficients in the reference MODEL's two arrays instead of their concatenation.  shs_rest != NULL: `shs` holds the DC coefficient only
     * (P x 1 x 3, `_features_dc`) and `shs_rest` the other M - 1 (P x (M - 1) x 3, `_features_rest`); M stays the total count.  Saves the caller the
     * torch.cat of scene/gaussian_model.py:140-143 (192 B read + written per Gaussian and call at M = 16, and its mirror image in the backward).  Both 16-byte aligned.
     * Results are bit-identical to the concatenated form. */
    const float* shs_rest;
    /* IBGS_FLAG_SRC_DEPTH_SLOTS (round 6): `src_depths` is a TABLE of depth planes (any number of H x W planes, e.g. the trainer's depth cache
     * scene.rendered_depth_list: one plane per training camera) and source m's plane is number src_depth_slot[m] of it -- what the reference gets by indexing,
     * rendered_depth_list[src_idx] (gaussian_renderer/__init__.py:255), i.e. by a copy of n_src planes per call.  Without the flag the field is ignored. */
    int32_t src_depth_slot[IBGS_MAX_SRC];
} ibgs_forward_args;

typedef struct ibgs_backward_args {
    void* stream;
    int32_t P, D, M, W, H;
    int64_t R;                   /* value returned by ibgs_forward */
    const float* means3D;
    const float* shs;
    const float* colors_precomp;
    const float* scales;
    const float* rotations;
    const float* cov3D_precomp;
    const float* all_map;
    float scale_modifier;
    const float* bg;
    const float* viewmatrix;
    const float* projmatrix;
    const float* campos;
    float tanfovx, tanfovy;
    int32_t n_src;
    const float* ref_to_src;
    const float* src_cam_pos;
    const float* src_images;
    const float* src_depths;
    const int32_t* radii;
    /* forward results that the kernels re-read (backward.cu:705, 734) */
    const float* out_depth;   /* median depth, 1 x H x W */
    const float* out_warped;  /* 15 x H x W */
    /* arenas exactly as the forward left them */
    char* geom; char* binning; char* img;
    char* tex; size_t tex_bytes;   /* render_geo only: transient scratch, >= ibgs_required_tex(n_src, W, H) */
    /* incoming gradients (NULL = zero) */
    const float* dL_dcolor;   /* 3 x H x W */
    const float* dL_dnormal;  /* 3 x H x W */
    const float* dL_ddepth;   /* 1 x H x W */
    const float* dL_dwarped;  /* 15 x H x W   (both NULL: the median / warp window pass of the backward is not run and `tex` / `geo_table` are not touched -- same
                                 gradients, bit for bit under IBGS_FLAG_DETERMINISTIC, as zero-filled arrays) */
    /* scratch: P x 16 floats, ZEROED by the caller (per-Gaussian accumulation rows) */
    float* grad_acc;
    /* gradient outputs: fully overwritten (dL_dscale / dL_drot only when scales is given, dL_dsh only when shs is) */
    float* dL_dmean2D;     /* P x 3 */
    float* dL_dmean2D_abs; /* P x 3 */
    float* dL_dconic;      /* P x 4 (x,y,w used); optional (may be NULL).  For a near-singular conic under the default summation (see IBGS_FLAG_REF_ARITH) it is derived back from dL/dcov2D */
    float* dL_dopacity;    /* P */
    float* dL_dcolors;     /* P x 3; may be NULL when shs is given without IBGS_FLAG_SH_FACTORED (it is the gradient of colors_precomp) */
    float* dL_dmean3D;     /* P x 3 */
    float* dL_dcov3D;      /* P x 6; may be NULL when scales + rotations are given (it is the gradient of cov3D_precomp) */
    float* dL_dsh;         /* P x M x 3 */
    float* dL_dscale;      /* P x 3 */
    float* dL_drot;        /* P x 4 */
    float* dL_dall_map;    /* P x 5 */
    int32_t render_geo;
    uint32_t flags;
    /* fused plane-map glue (same meaning as in ibgs_forward_args).  With plane_mode != 0, dL_dall_map may be NULL and
     * the gradient continues to the raw parameters: dL_dplane_normal (P x 3) and dL_dplane_offset (P, may be NULL) for
     * IBGS_PLANE_LEARNT; for IBGS_PLANE_SMALLEST_AXIS it is added into dL_drot.  Both modes add the d(distance)/d(mean)
     * term into dL_dmean3D. */
    const float* plane_normal;
    const float* plane_offset;
    int32_t plane_mode;
    float* dL_dplane_normal;
    float* dL_dplane_offset;
    /* render_geo: transient scratch of >= ibgs_required_geo_table(W, H) bytes -- the per-pixel table of median / warp terms that
     * the pixel-parallel window pass leaves for the blend loop (6 words per median buffer slot) */
    char* geo_table; size_t geo_table_bytes;
    /* IBGS_FLAG_DETERMINISTIC: transient scratch of >= ibgs_required_deterministic(R, P) bytes (slab R x 16 floats + sort buffers) */
    char* det_scratch; size_t det_scratch_bytes;
    /* the forward's buffer_length (0 = not stated).  When stated, geo_table only needs ibgs_required_geo_table_for(W, H, buffer_length)
     * bytes and det_scratch ibgs_required_deterministic_for(R, P, W, H, render_geo, flags).  It must not be SMALLER than the buffer_length
     * of the ibgs_forward whose arenas are handed in: the window pass then keeps only the first buffer_length buffered contributors of every
     * pixel (wrong median / warp gradients, but never a write past geo_table) */
    int32_t buffer_length;
    /* optional (may be NULL): ibgs_tile_order_slots(W, H) words that receive the order in which a colour backward launched its tiles, i.e.
     * what ibgs_forward_args.tile_order_hint of the same camera's next forward wants (the same words also land in the image arena) */
    uint32_t* tile_order_out;
    /* SH coefficients in two arrays (see ibgs_forward_args.shs_rest): then dL_dsh receives the DC gradient (P x 1 x 3) and dL_dsh_rest the rest
     * (P x (M - 1) x 3); both fully overwritten.  With IBGS_FLAG_SH_FACTORED neither is written. */
    const float* shs_rest;
    float* dL_dsh_rest;
} ibgs_backward_args;

size_t ibgs_required_deterministic(int64_t R, int32_t P);          /* any frame, any flags: four rows per list entry */
size_t ibgs_required_deterministic_for(int64_t R, int32_t P, int32_t W, int32_t H, int32_t render_geo, uint32_t flags);   /* rows = the waves per tile
                                                                      the backward will use for this frame (1 from 4096 tiles on): a quarter at 1080p */
size_t ibgs_required_geo_table(int32_t W, int32_t H);              /* any buffer_length (8 slots) */
size_t ibgs_required_geo_table_for(int32_t W, int32_t H, int32_t buffer_length);   /* buffer_length + 1 slots: 250 MB instead of 400 MB at 1080p, L = 4 */
size_t ibgs_required_geom(int32_t P);
size_t ibgs_required_img(int32_t W, int32_t H);
size_t ibgs_required_binning(int64_t R, int32_t W, int32_t H);
size_t ibgs_required_tex(int32_t n_src, int32_t W, int32_t H);

/* words of a tile order (the tiles of the frame rounded up to a multiple of 1024) */
size_t ibgs_tile_order_slots(int32_t W, int32_t H);
int64_t ibgs_forward(const ibgs_forward_args* args);
/* diagnostics of the calling thread's last ibgs_forward: out[0] = R, out[1] = coarse binning entries (-1 unless rendered_hint was used),
 * out[2] = 1 when the hint was too small and binning + render ran a second time with the exact size */
void ibgs_last_forward_stats(int64_t* out3);
/* The one asynchronous error of the library: a forward with a rendered_hint does not wait for its depth sort, whose decoupled look-back gives up (and leaves mis-ordered
 * lists) if a workgroup is starved for seconds.  The sticky error word is reported by the ibgs_backward of the same step (it waits briefly for the word of its forward, with
 * its own kernels already queued: no optimiser step sees such gradients), else by the next ibgs_forward on the stream, else HERE -- for callers no backward follows
 * (evaluation renders, the last forward of a run).  wait != 0 drains the stream first (the answer is then final); returns 0 or -IBGS_ERR_HIP (ibgs_last_error). */
int32_t ibgs_check_async(void* stream, int32_t wait);
int32_t ibgs_backward(const ibgs_backward_args* args);
int32_t ibgs_mark_visible(void* stream, int32_t P, const float* means3D, const float* viewmatrix,
                          const float* projmatrix, uint8_t* present /* P bools */);

/* View-parallel step, SURVEY 8(e): dL_dsh[i][k][c] = sum over views v of basis_k(normalise(means3D[i] - camposes[v]))
 * * dcolor[v][i][c] for k < (D+1)^2, zero for the other coefficients; dL_dsh (P x M x 3) is fully overwritten.
 * `dcolor` = the dL_dcolors outputs of n_views ibgs_backward calls made with IBGS_FLAG_SH_FACTORED, stacked
 * (n_views x P x 3, view v starting at dcolor + v * view_stride floats; view_stride = 0 means P * 3); `camposes` = their
 * camera centres (n_views x 3).  A stride > P * 3 lets an exchange carry per-view extras (e.g. the camera centre) behind
 * each view's block in ONE all-gather.  No reference counterpart (single GPU). */
int32_t ibgs_sh_grad_from_views(void* stream, int32_t P, int32_t D, int32_t M, int32_t n_views, const float* means3D,
                                const float* camposes, const float* dcolor, int64_t view_stride, float* dL_dsh);

/* Section 8(f) "next" row 4 -- the trainer's optimiser step (train.py:421-430: torch.optim.Adam over the eight Gaussian
 * parameter groups of scene/gaussian_model.py:227-241) as ONE launch.  Update rule and operation order of
 * torch.optim.Adam without weight decay / amsgrad; bias_correction{1,2} = 1 - beta{1,2}^step are computed by the caller.
 * All pointers are device fp32 arrays of `numel` elements, updated in place (param, exp_avg, exp_avg_sq). */
#define IBGS_ADAM_MAX_TENSORS 16
typedef struct ibgs_adam_tensor {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t numel;
    double lr, beta1, beta2, eps;                  /* as the optimiser holds them (Python floats); 1 - beta is formed in double */
    double bias_correction1, bias_correction2;      /* 1 - beta1^step, 1 - beta2^step */
} ibgs_adam_tensor;
int32_t ibgs_adam_step(void* stream, int32_t n_tensors, const ibgs_adam_tensor* tensors);
/* ... and the SH coefficients straight from the per-view factors of IBGS_FLAG_SH_FACTORED backwards (round 6): what ibgs_sh_grad_from_views followed by
 * ibgs_adam_step computes for them -- bit for bit --, without the dense P x M x 3 gradient ever being written or read (192 of the ~285 bytes per Gaussian that
 * ibgs_backward's per-Gaussian kernel writes, and 1/7 of the optimiser step's traffic on these tensors).  means3D, camposes, dcolor, view_stride, D, n_views as for
 * ibgs_sh_grad_from_views.  1 or 2 tensors: tensor t holds coefficients first_coeff[t] .. first_coeff[t] + n_coeff[t] - 1 of every Gaussian (P x n_coeff[t] x 3
 * floats; the reference's `_features_dc` = {0, 1} and `_features_rest` = {1, 15}, each with its own learning rate; a combined P x 16 x 3 tensor = {0, 16});
 * `grad` of the tensors is ignored.  Coefficients above the active degree D receive a zero gradient (their moments still decay, as under torch.optim.Adam).
 * No reference counterpart: train.py runs torch.optim.Adam on the dense gradient. */
int32_t ibgs_adam_step_sh(void* stream, int32_t P, int32_t D, int32_t n_views, const float* means3D, const float* camposes, const float* dcolor,
                          int64_t view_stride, int32_t n_tensors, const ibgs_adam_tensor* tensors, const int32_t* first_coeff, const int32_t* n_coeff);

/* The photometric L1 term of the trainer's loss, l1_loss(image, gt) = |image - gt|.mean() (utils/loss_utils.py:23-24, train.py:302),
 * value and gradient in ONE pass over the image: *loss = mean |x - y|, grad[i] = sign(x[i] - y[i]) / n (grad may be NULL: value only).
 * x, y, grad: n floats each on the device; loss: one device float; `scratch` >= ibgs_required_l1() bytes, caller-owned, transient.
 * No float atomics: the partial sums are added in a fixed order. */
size_t ibgs_required_l1(void);
int32_t ibgs_l1_loss(void* stream, int64_t n, const float* x, const float* y, float* grad, float* loss, char* scratch, size_t scratch_bytes);
/* The gradient alone, scaled by a DEVICE scalar (autograd's incoming gradient; NULL = 1): grad[i] = sign(x[i] - y[i]) * (*scale_dev) / n. */
int32_t ibgs_l1_grad(void* stream, int64_t n, const float* x, const float* y, const float* scale_dev, float* grad);
/* ... or, when ibgs_l1_loss already stored grad = sign(x - y) / n: grad[i] *= *scale_dev in place.  A scale of exactly 1.0 (a loss term that is
 * added to the total unweighted) is detected on the device and moves no data: value + gradient of the term then cost ONE pass over x and y. */
int32_t ibgs_l1_rescale(void* stream, int64_t n, float* grad, const float* scale_dev);

/* Row G(vii) of SURVEY 8(a): the depth -> normal map of the render glue (utils/graphics_utils.py:17-83 `normal_from_depth_image`, offset = None, through
 * gaussian_renderer/__init__.py:16-26 `render_normal` at scale 1, and the normalisation render() applies to it, :338-342), one kernel each way.
 * depth: H x W; normal / dL_dnormal: 3 x H x W (planar); dL_ddepth: H x W, every element written.  fx, fy, cx, cy: the pinhole intrinsics of
 * Camera.get_calib_matrix_nerf (scene/cameras.py:118-121).  The border pixels' normals are zero, as the reference pads them. */
int32_t ibgs_depth_normal_forward(void* stream, int32_t W, int32_t H, float fx, float fy, float cx, float cy, const float* depth, float* normal);
int32_t ibgs_depth_normal_backward(void* stream, int32_t W, int32_t H, float fx, float fy, float cx, float cy, const float* depth,
                                   const float* dL_dnormal, float* dL_ddepth);

/* Row G of SURVEY 8(a), model side: the activations render() applies to the raw parameters on every call (scene/gaussian_model.py:44-52, 128-147) -- scale =
 * exp(raw_scale) (P x 3), rot = F.normalize(raw_rot) = raw_rot / max(|raw_rot|, 1e-12) (P x 4), opacity = sigmoid(raw_opacity) (P x 1) -- one kernel each way
 * instead of ~17 torch launches.  Any of the three may be left out (NULL input; backward: NULL output).  Backward: d_* = dL/d raw_* from g_* = dL/d activated,
 * every element written.  Arithmetic as torch's kernels, operation by operation (expf, 1 / (1 + expf(-x)), sqrtf of the sum of squares). */
int32_t ibgs_activate_forward(void* stream, int32_t P, const float* raw_scale, const float* raw_rot, const float* raw_opacity, float* scale, float* rot, float* opacity);
int32_t ibgs_activate_backward(void* stream, int32_t P, const float* raw_scale, const float* raw_rot, const float* raw_opacity,
                               const float* g_scale, const float* g_rot, const float* g_opacity, float* d_scale, float* d_rot, float* d_opacity);

/* Section 8(f) "next" row 3 -- replaces simple_knn._C.distCUDA2 (submodules/simple-knn/spatial.cu:15-26,
 * simple_knn.cu:185-220): out[i] = mean of the three smallest squared distances from point i to the other
 * points (exact).  `scratch` >= ibgs_required_knn(P) bytes, caller-owned, transient. */
size_t ibgs_required_knn(int32_t P);
int32_t ibgs_knn_mean_dist2(void* stream, int32_t P, const float* points /* P x 3 */, float* out /* P */,
                            char* scratch, size_t scratch_bytes);

/* SURVEY 8(f) row 4 -- densification surgery as one data-movement pass.  Replaces the per-tensor boolean indexing and
 * torch.cat of the reference's _prune_optimizer / cat_tensors_to_optimizer / prune_points (scene/gaussian_model.py:377-444):
 *   dst[0 .. n_keep)              = the rows of src whose keep_mask byte is non-zero, in their original order
 *   dst[n_keep .. n_keep + n_app) = the rows of `append` (copied), or zeros when `append` is NULL (Adam moments of new points)
 * for EVERY tensor of the call (parameters, exp_avg, exp_avg_sq, per-point statistics ...) in one launch.
 * ibgs_compact_plan scans the mask into `scratch` (>= ibgs_required_compact(n_old) bytes, caller-owned, must stay alive until
 * ibgs_compact_apply has run) and returns n_keep -- the one host synchronisation, needed to size dst (the reference's boolean
 * indexing synchronises per tensor).  keep_mask NULL keeps every row.  Pure data movement: bit-identical to the torch form. */
#define IBGS_COMPACT_MAX_TENSORS 40
typedef struct ibgs_compact_tensor {
    const float* src;      /* n_old x width */
    const float* append;   /* n_app x width, or NULL: appended rows are zeros */
    float* dst;            /* (n_keep + n_app) x width, caller-allocated */
    int32_t width;         /* 4-byte elements per row (any 4-byte type travels as float bits) */
    int32_t reserved;
} ibgs_compact_tensor;
size_t ibgs_required_compact(int32_t n_old);
int64_t ibgs_compact_plan(void* stream, int32_t n_old, const uint8_t* keep_mask /* n_old or NULL */, char* scratch, size_t scratch_bytes);
int32_t ibgs_compact_apply(void* stream, int32_t n_tensors, const ibgs_compact_tensor* tensors, int32_t n_old, int32_t n_app,
                           const char* scratch);
/* The densification statistics of one training view (train.py:400-405; GaussianModel.add_densification_stats, scene/gaussian_model.py:600-604) in one
 * launch and without the reference's boolean-index host syncs: for every Gaussian with radii[i] > 0: max_radii2D[i] = max(max_radii2D[i], radii[i]),
 * accum[i] += |dL_dmean2D[i].xy|, accum_abs[i] += |dL_dmean2D_abs[i].xy|, denom[i] += 1, denom_abs[i] += 1.  dL_dmean2D*: P x 3 (the `.grad` of the
 * viewspace sinks); accum* / denom*: P (or P x 1) floats; any output (and its input) may be NULL. */
int32_t ibgs_densify_stats(void* stream, int32_t P, const int32_t* radii, const float* dL_dmean2D, const float* dL_dmean2D_abs,
                           float* accum, float* accum_abs, float* denom, float* denom_abs, float* max_radii2D);

/* Introspection for tests: byte offsets of the named sub-arrays inside the arenas.
 * Returns -1 for an unknown name. Names: see DESIGN.md section 2. */
int64_t ibgs_geom_offset(int32_t P, const char* name);
int64_t ibgs_img_offset(int32_t W, int32_t H, const char* name);
int64_t ibgs_binning_offset(int64_t R, int32_t W, int32_t H, const char* name);

/* Optional per-stage timing for bench.py's roofline object: when enabled, ibgs_forward / ibgs_backward
 * bracket the selected stages with hipEvents on the caller's stream.  ibgs_timing_collect synchronises
 * those events and returns, per stage, the summed milliseconds and the number of launches since the last
 * collect.  The only library state besides the last-error string; off by default. */
#define IBGS_STAGE_PREPROCESS 0
#define IBGS_STAGE_DEPTH_SORT 1
#define IBGS_STAGE_SCAN 2
#define IBGS_STAGE_EMIT 3         /* two-level binning, part 1: coarse entries placed per cell in depth order, per-tile counts, ranges */
#define IBGS_STAGE_TILE_SORT 4    /* two-level binning, part 2: Gaussian ids to their list slots */
#define IBGS_STAGE_RANGES 5       /* (no kernel of its own any more) */
#define IBGS_STAGE_RENDER_FWD 6
#define IBGS_STAGE_RENDER_BWD 7    /* the backward's blend kernel alone (render_bwd_color_kernel / render_bwd_geo*_kernel) */
#define IBGS_STAGE_PREPROCESS_BWD 8
#define IBGS_STAGE_GEO_WINDOW 9    /* render_geo backward: the pixel-parallel window pass in front of the blend kernel */
#define IBGS_STAGE_TILE_ORDER 10   /* the backward's launch-order kernel (+ the deterministic mode's slab reduction) */
#define IBGS_NUM_STAGES 11
void ibgs_timing_enable(uint32_t stage_mask);   /* bit i = time stage i; 0 disables */
int32_t ibgs_timing_collect(float* ms /* IBGS_NUM_STAGES */, int32_t* launches /* IBGS_NUM_STAGES */);

/* ABI self-check for FFI bindings: sizeof the two argument structs as this library was built. */
size_t ibgs_sizeof_forward_args(void);
size_t ibgs_sizeof_backward_args(void);

const char* ibgs_last_error(void);
const char* ibgs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* IBGS_RAST_H */
