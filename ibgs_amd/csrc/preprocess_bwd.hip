// B3 + B4: per-Gaussian backward -- conic -> 3D covariance / mean, 2D mean -> 3D mean, RGB -> SH,
// 3D covariance -> scale / rotation.
//
// Behaviour: DPR/cuda_rasterizer/backward.cu:241-371 (computeCov2DCUDA), 443-493 (preprocessCUDA),
// 116-235 (computeColorFromSH backward), 375-438 (computeCov3D backward).  The reference runs two
// kernels and reads gradients from five atomically-accumulated arrays; here one fused kernel reads
// the 64-byte accumulation row written by render_bwd.hip (one coalesced line per Gaussian) and writes
// every output of the operator in the reference's layout (rasterize_points.cu:209-219).
#include "common.h"
#include "adam_math.h"          // AdamDev, adam_one (adam_sh_kernel below)

namespace ibgs {

__constant__ float bC1 = 0.4886025119029199f;
__constant__ float bC0 = 0.28209479177387814f;
__constant__ float bC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                             -1.0925484305920792f, 0.5462742152960396f};
__constant__ float bC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                             0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                             -0.5900435899266435f};

struct PreBwdParams {
    int P, D, M;
    const float* means3D; const int32_t* radii; const float* shs; const float* shs_rest; const uint8_t* clamped;          // shs_rest: see ibgs_backward_args
    const float* scales; const float* rotations; float scale_modifier;
    const float* cov3D;        // precomputed input or the forward's computed one
    float* gacc;               // P x 16 moment rows written by render_bwd.hip (re-zeroed here when clear_gacc)
    int clear_gacc;
    int risky_rows;            // format of a near-singular conic's row (render_bwd.hip): 0 = moments like every other row (no reference branch), 1 = RA_LFORM sums, 2 = RA_ASSOC (the reference's sums)
    const float* rec;          // P x 16 forward records (conic, opacity)
    float* dL_dmean2D; float* dL_dmean2D_abs; float* dL_dconic; float* dL_dopacity; float* dL_dcolors;
    float* dL_dall_map; float* dL_dmean3D; float* dL_dcov3D; float* dL_dsh; float* dL_dsh_rest; float* dL_dscale; float* dL_drot;
    const float* plane_normal; const float* plane_offset; int plane_mode;     // fused plane-map glue (common.h)
    float* dL_dplane_normal; float* dL_dplane_offset;
};

// FAST16 (M == 16, the layout of max SH degree 3): one wave per workgroup; the 192-B SH rows are read
// with 16-B vector loads and dL/dsh leaves through an LDS transpose so that the wave stores its contiguous
// 12 KB block with fully coalesced 16-B writes (a per-lane row store is 48 instructions that each touch 64
// different cache lines).
// Real SH basis values for the unit direction (x, y, z), degree 0..D (forward.cu:20-70 / backward.cu:114-160
// use the same polynomials); returns how many of the 16 entries are active.
__device__ __forceinline__ int sh_basis(int D, float x, float y, float z, float (&B)[16])
{
    int nb = 1;
    B[0] = bC0;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    if (D > 0) {
        B[1] = -bC1 * y; B[2] = bC1 * z; B[3] = -bC1 * x; nb = 4;
        if (D > 1) {
            B[4] = bC2[0] * xy; B[5] = bC2[1] * yz; B[6] = bC2[2] * (2.0f * zz - xx - yy);
            B[7] = bC2[3] * xz; B[8] = bC2[4] * (xx - yy); nb = 9;
            if (D > 2) {
                B[9] = bC3[0] * y * (3.0f * xx - yy); B[10] = bC3[1] * xy * z;
                B[11] = bC3[2] * y * (4.0f * zz - xx - yy);
                B[12] = bC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                B[13] = bC3[4] * x * (4.0f * zz - xx - yy); B[14] = bC3[5] * z * (xx - yy);
                B[15] = bC3[6] * x * (xx - 3.0f * yy); nb = 16;
            }
        }
    }
    return nb;
}

// WRITE_SH = false is the view-parallel "factored" mode (IBGS_FLAG_SH_FACTORED): dL/dsh of one view is the
// outer product basis(dir) x dL/dRGB, so only the clamp-masked dL/dRGB (3 floats instead of 3 M) leaves this
// kernel, in dL_dcolors; ibgs_sh_grad_from_views (below) rebuilds the summed dL/dsh after the exchange.
template <bool FAST16, bool WRITE_SH, bool SPLIT = false>          // SPLIT: DC and rest coefficients in two arrays (ibgs_backward_args.shs_rest), its own instantiations
__global__ void __launch_bounds__(FAST16 ? 64 : 256) preprocess_bwd_kernel(PreBwdParams p, Cam cam)
{
    __shared__ float4 s_t[FAST16 ? 64 * 13 : 1];     // row stride 13 quads: conflict-free b128 access
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < p.P;
    bool vis = valid && (p.radii[i] > 0);
    // A visible Gaussian that no pixel blended (everything in front of it was already opaque, or it fell off every tile list)
    // has an all-zero moment row, hence all-zero gradients: it is handled like an invisible one and its 200+ bytes of SH,
    // covariance and record reads are skipped.  Half of the Gaussians of the C3 bench scene are in that class.
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0, g3 = g0;
    if (vis) {
        float4* grow = reinterpret_cast<float4*>(p.gacc + (size_t)i * GACC_FLOATS);
        g0 = grow[0]; g1 = grow[1]; g2 = grow[2]; g3 = grow[3];
        const uint32_t any = (__float_as_uint(g0.x) | __float_as_uint(g0.y) | __float_as_uint(g0.z) | __float_as_uint(g0.w) |
                              __float_as_uint(g1.x) | __float_as_uint(g1.y) | __float_as_uint(g1.z) | __float_as_uint(g1.w) |
                              __float_as_uint(g2.x) | __float_as_uint(g2.y) | __float_as_uint(g2.z) | __float_as_uint(g2.w) |
                              __float_as_uint(g3.x) | __float_as_uint(g3.y) | __float_as_uint(g3.z) | __float_as_uint(g3.w)) & 0x7FFFFFFFu;
        if (any == 0u) vis = false;
        else if (p.clear_gacc) { const float4 z = make_float4(0.f, 0.f, 0.f, 0.f); grow[0] = z; grow[1] = z; grow[2] = z; grow[3] = z; }
    }
    float gv[(FAST16 && WRITE_SH) ? 48 : 1];
    if (FAST16 && WRITE_SH) {
#pragma unroll
        for (int k = 0; k < 48; k++) gv[k] = 0.f;
    }
    if (valid && !vis) {
        // invisible (or untouched) Gaussian: every gradient is zero; written explicitly so that callers need no memset
        p.dL_dmean2D[3 * i] = 0.f; p.dL_dmean2D[3 * i + 1] = 0.f; p.dL_dmean2D[3 * i + 2] = 0.f;
        if (p.dL_dmean2D_abs) { p.dL_dmean2D_abs[3 * i] = 0.f; p.dL_dmean2D_abs[3 * i + 1] = 0.f; p.dL_dmean2D_abs[3 * i + 2] = 0.f; }          // (NULL: IBGS_FLAG_NO_ABS_GRAD)
        if (p.dL_dconic) { p.dL_dconic[4 * i] = 0.f; p.dL_dconic[4 * i + 1] = 0.f; p.dL_dconic[4 * i + 2] = 0.f; p.dL_dconic[4 * i + 3] = 0.f; }
        p.dL_dopacity[i] = 0.f;
        if (p.dL_dcolors) { p.dL_dcolors[3 * i] = 0.f; p.dL_dcolors[3 * i + 1] = 0.f; p.dL_dcolors[3 * i + 2] = 0.f; }
        if (p.dL_dall_map) for (int k = 0; k < 5; k++) p.dL_dall_map[5 * i + k] = 0.f;
        if (p.dL_dplane_normal) { p.dL_dplane_normal[3 * i] = 0.f; p.dL_dplane_normal[3 * i + 1] = 0.f; p.dL_dplane_normal[3 * i + 2] = 0.f; }
        if (p.dL_dplane_offset) p.dL_dplane_offset[i] = 0.f;
        p.dL_dmean3D[3 * i] = 0.f; p.dL_dmean3D[3 * i + 1] = 0.f; p.dL_dmean3D[3 * i + 2] = 0.f;
        if (p.dL_dcov3D) for (int k = 0; k < 6; k++) p.dL_dcov3D[6 * i + k] = 0.f;
        if (!FAST16 && WRITE_SH && p.shs) {
            if (SPLIT) {
                float* gd = p.dL_dsh + (size_t)i * 3; float* gr = p.dL_dsh_rest + (size_t)i * (p.M - 1) * 3;
                gd[0] = gd[1] = gd[2] = 0.f;
                for (int k = 0; k < 3 * (p.M - 1); k++) gr[k] = 0.f;
            } else { float* gsh = p.dL_dsh + (size_t)i * p.M * 3; for (int k = 0; k < 3 * p.M; k++) gsh[k] = 0.f; }
        }
        if (p.scales) {
            p.dL_dscale[3 * i] = 0.f; p.dL_dscale[3 * i + 1] = 0.f; p.dL_dscale[3 * i + 2] = 0.f;
            p.dL_drot[4 * i] = 0.f; p.dL_drot[4 * i + 1] = 0.f; p.dL_drot[4 * i + 2] = 0.f; p.dL_drot[4 * i + 3] = 0.f;
        }
    }
    if (FAST16 && SPLIT && p.shs) {
        // DC and the rest in two arrays (ibgs_backward_args.shs_rest): the wave's 64 rows are one contiguous block per array (720 + 48 float4 = 12 x 64
        // pieces), fetched with coalesced 16-B loads -- pieces that hold nothing of a Gaussian that needs its coefficients are skipped -- and parked in the
        // transpose buffer AS THEY LIE (16-byte stores); each lane then reads its own row word by word: rows 45 (3) words apart, odd strides, no bank
        // conflicts.  (The combined layout's per-lane 16-B loads need 16-byte aligned rows; 180-byte rows are not.)
        const int lane = threadIdx.x;
        const uint64_t vism = __builtin_amdgcn_ballot_w64(vis);
        const int i0 = blockIdx.x * 64;
        const int nrows = min(64, p.P - i0);          // (the last block may be partial: its pieces end with the arrays; lanes beyond them are not `vis`)
        const float4* rest4 = reinterpret_cast<const float4*>(p.shs_rest + (size_t)i0 * 45);
        const float4* dc4 = reinterpret_cast<const float4*>(p.shs + (size_t)i0 * 3);
        float4 v[12];
#pragma unroll
        for (int it = 0; it < 12; it++) {
            const int q = it * 64 + lane;
            v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < 720) {
                const int ra = (4 * q) / 45, rb = (4 * q + 3) / 45;
                if (((vism >> ra) & 1ull) || ((vism >> rb) & 1ull)) {
                    if (4 * q + 3 < nrows * 45) v[it] = rest4[q];
                    else { const float* sp = p.shs_rest + (size_t)i0 * 45; float* f = &v[it].x; for (int k = 0; k < 4; k++) if (4 * q + k < nrows * 45) f[k] = sp[4 * q + k]; }
                }
            } else {
                const int qd = q - 720, ra = (4 * qd) / 3, rb = (4 * qd + 3) / 3;
                if ((((vism >> ra) & 3ull) != 0ull) || ((vism >> rb) & 1ull)) {
                    if (4 * qd + 3 < nrows * 3) v[it] = dc4[qd];
                    else { const float* sp = p.shs + (size_t)i0 * 3; float* f = &v[it].x; for (int k = 0; k < 4; k++) if (4 * qd + k < nrows * 3) f[k] = sp[4 * qd + k]; }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 12; it++) s_t[it * 64 + lane] = v[it];          // rest: words 0 .. 2879, DC: words 2880 .. 3071
        __syncthreads();
    }
    if (vis) {

    const float4* rrow = reinterpret_cast<const float4*>(p.rec + (size_t)i * REC_FLOATS);
    const float4 r0 = rrow[0], r1 = rrow[1];
    // Moments of q = o*G*dL/dalpha over all pixels (render_bwd.hip) -> the reference's quantities
    // (backward.cu:786-804): dG/ddelx = -G (a dx + b dy), conic grads -0.5 G d d^T, dL/do = G dL/dalpha.
    const float ddelx_dx = (float)(0.5 * cam.W), ddely_dy = (float)(0.5 * cam.H);
    const float ca2 = r1.x, cb2 = r1.y, cc2 = r1.z, opa = r0.z;
    // A near-singular conic's row (render_bwd.hip, "near-singular conics"): RA_LFORM -- sums of q l, |q l|, q l l^T with l = (conic in exp2 units) d: the conic is
    // already applied, and the three second moments ARE dL/dcov2D up to 0.5, 1, 0.5 (and the scale): taken below instead of the chain of backward.cu:405-420;
    // RA_ASSOC -- the reference's own per-pair quantities (dL/dG dG/ddel, its magnitude, gd d dL/dG, G dL/dalpha): only 0.5 W, 0.5 H and -0.5 remain to be applied
    const int risky_row = conic_takes_ref_power(ca2, cb2, cc2) ? p.risky_rows : 0;
    const bool ref_row = risky_row == 2, l_row = risky_row == 1;
    const float g2x = ref_row ? ddelx_dx * g0.x : (l_row ? -ddelx_dx * (g0.x * EXP2_UNSCALE) : -ddelx_dx * (ca2 * g0.x + cb2 * g0.y));   // dL/dmean2D
    const float g2y = ref_row ? ddely_dy * g0.y : (l_row ? -ddely_dy * (g0.y * EXP2_UNSCALE) : -ddely_dy * (cc2 * g0.y + cb2 * g0.x));
    const float gcx = -0.5f * g1.x, gcy = -0.5f * g1.y, gcz = -0.5f * g1.z; // dL/dconic (a, b, c)   (l_row: not the conic gradient -- see dL_dconic below)
    const float gcol[3] = {g2.x, g2.y, g2.z};

    p.dL_dmean2D[3 * i] = g2x; p.dL_dmean2D[3 * i + 1] = g2y; p.dL_dmean2D[3 * i + 2] = 0.f;
    // (the blend kernels form sum |q (conic d)| with the conic in exp2 units, common.h: undone here, once per Gaussian)
    const float abs_unscale = ref_row ? 1.0f : EXP2_UNSCALE;
    if (p.dL_dmean2D_abs) { p.dL_dmean2D_abs[3 * i] = ddelx_dx * (g0.z * abs_unscale); p.dL_dmean2D_abs[3 * i + 1] = ddely_dy * (g0.w * abs_unscale); p.dL_dmean2D_abs[3 * i + 2] = 0.f; }
    if (p.dL_dconic && !l_row) { p.dL_dconic[4 * i] = gcx; p.dL_dconic[4 * i + 1] = gcy; p.dL_dconic[4 * i + 2] = 0.f; p.dL_dconic[4 * i + 3] = gcz; }          // (l_row: below, once cov2D is known)
    p.dL_dopacity[i] = ref_row ? g1.w : (opa > 0.f ? g1.w / opa : 0.f);          // (l_row: sum q, like every other row)
    if ((WRITE_SH || !p.shs) && p.dL_dcolors) { p.dL_dcolors[3 * i] = gcol[0]; p.dL_dcolors[3 * i + 1] = gcol[1]; p.dL_dcolors[3 * i + 2] = gcol[2]; }
    if (p.dL_dall_map) {
        p.dL_dall_map[5 * i] = g2.w; p.dL_dall_map[5 * i + 1] = g3.x; p.dL_dall_map[5 * i + 2] = g3.y;
        p.dL_dall_map[5 * i + 3] = 0.f; p.dL_dall_map[5 * i + 4] = g3.z;
    }

    const float* __restrict__ vm = cam.vm; const float* __restrict__ pm = cam.pm;
    const float mean[3] = {p.means3D[3 * i], p.means3D[3 * i + 1], p.means3D[3 * i + 2]};

    // ---- plane-map glue backward: (dL/dn_cam, dL/ddist) -> raw normal / offset (mode 1) or rotation (mode 2), + mean ----
    float pgm[3] = {0.f, 0.f, 0.f}, prot[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.plane_mode) {
        const PlaneEval e = plane_eval(p.plane_mode, p.plane_normal, p.plane_offset, p.scales, p.rotations, i, mean[0], mean[1], mean[2], cam.campos, vm);
        const float gnc[3] = {g2.w, g3.x, g3.y};
        const float ggd = e.sgn * g3.z;                                   // dL/d(signed world-frame plane distance)
        float gl[3], gn0[3];
#pragma unroll
        for (int j = 0; j < 3; j++) gl[j] = gnc[j] - ggd * vm[12 + j];     // d_cam = gd - n_cam . V[3,:3]
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float gn = gl[0] * vm[4 * k] + gl[1] * vm[4 * k + 1] + gl[2] * vm[4 * k + 2] - ggd * mean[k];   // n_cam = n @ V[:3,:3]; gd = -(n . x)
            gn0[k] = e.flip * gn;                                          // back through the flip
            pgm[k] = -ggd * e.n[k];
        }
        if (p.plane_mode == IBGS_PLANE_LEARNT) {
            const float u[3] = {e.n[0] * e.flip, e.n[1] * e.flip, e.n[2] * e.flip};           // raw / |raw|
            const float dot = u[0] * gn0[0] + u[1] * gn0[1] + u[2] * gn0[2];
#pragma unroll
            for (int k = 0; k < 3; k++) p.dL_dplane_normal[3 * i + k] = (gn0[k] - u[k] * dot) * e.inv_len;
            if (p.dL_dplane_offset) p.dL_dplane_offset[i] = e.flip * ggd;
        } else {
            // column e.axis of R(q) w.r.t. the given quaternion (r, x, y, z); any component along q is removed by the
            // caller's normalisation backward, so the unit-quaternion form of R is differentiated as is
            const float r_ = p.rotations[4 * i], x = p.rotations[4 * i + 1], y = p.rotations[4 * i + 2], z = p.rotations[4 * i + 3];
            const float a = gn0[0], b = gn0[1], c = gn0[2];
            if (e.axis == 0) {
                prot[0] = 2.f * (z * b - y * c); prot[1] = 2.f * (y * b + z * c);
                prot[2] = -4.f * y * a + 2.f * x * b - 2.f * r_ * c; prot[3] = -4.f * z * a + 2.f * r_ * b + 2.f * x * c;
            } else if (e.axis == 1) {
                prot[0] = 2.f * (x * c - z * a); prot[1] = 2.f * y * a - 4.f * x * b + 2.f * r_ * c;
                prot[2] = 2.f * (x * a + z * c); prot[3] = -2.f * r_ * a - 4.f * z * b + 2.f * y * c;
            } else {
                prot[0] = 2.f * (y * a - x * b); prot[1] = 2.f * z * a - 2.f * r_ * b - 4.f * x * c;
                prot[2] = 2.f * r_ * a + 2.f * z * b - 4.f * y * c; prot[3] = 2.f * (x * a + y * b);
            }
        }
    }
    float c6[6];
#pragma unroll
    for (int k = 0; k < 6; k++) c6[k] = p.cov3D[6 * i + k];

    // ---- cov2D backward ----
    float t[3] = {vm[0] * mean[0] + vm[4] * mean[1] + vm[8] * mean[2] + vm[12],
                  vm[1] * mean[0] + vm[5] * mean[1] + vm[9] * mean[2] + vm[13],
                  vm[2] * mean[0] + vm[6] * mean[1] + vm[10] * mean[2] + vm[14]};
    const float limx = 1.3f * cam.tanfovx, limy = 1.3f * cam.tanfovy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
    t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
    const float xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    const float ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float fx = cam.fx, fy = cam.fy;
    const float j00 = fx / t[2], j02 = -(fx * t[0]) / (t[2] * t[2]);
    const float j11 = fy / t[2], j12 = -(fy * t[1]) / (t[2] * t[2]);
    float A[2][3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float rv0 = vm[4 * r + 0], rv1 = vm[4 * r + 1], rv2 = vm[4 * r + 2];
        A[0][r] = rv0 * j00 + rv1 * 0.0f + rv2 * j02;
        A[1][r] = rv0 * 0.0f + rv1 * j11 + rv2 * j12;
    }
    const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float SA[2][3];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 3; r++) SA[a][r] = S[r][0] * A[a][0] + S[r][1] * A[a][1] + S[r][2] * A[a][2];
    const float a_ = A[0][0] * SA[0][0] + A[0][1] * SA[0][1] + A[0][2] * SA[0][2] + 0.3f;
    const float b_ = A[0][0] * SA[1][0] + A[0][1] * SA[1][1] + A[0][2] * SA[1][2];
    const float c_ = A[1][0] * SA[1][0] + A[1][1] * SA[1][1] + A[1][2] * SA[1][2] + 0.3f;
    const float denom = a_ * c_ - b_ * b_;
    float da = 0, db = 0, dc = 0;
    const float d2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float gS[6] = {0, 0, 0, 0, 0, 0};
    if (d2inv != 0) {
        da = d2inv * (-c_ * c_ * gcx + 2 * b_ * c_ * gcy + (denom - a_ * c_) * gcz);
        dc = d2inv * (-a_ * a_ * gcz + 2 * a_ * b_ * gcy + (denom - a_ * c_) * gcx);
        db = d2inv * 2 * (b_ * c_ * gcx - (denom + 2 * b_ * b_) * gcy + a_ * b_ * gcz);
        if (l_row) {
            // the three expressions above are -(c dx - b dy)^2, -(a dy - b dx)^2 and their product summed over the pairs, i.e. det^2 x the sums of q l_x^2, q l_y^2,
            // q l_x l_y with l = conic d -- which is what the blend accumulated for this Gaussian, pair by pair, with nothing cancelling (render_bwd.hip: RA_LFORM)
            const float kk = d2inv * (denom * denom);          // = det^2 / (det^2 + 1e-7): the reference's regulariser, kept
            const float m00 = g1.x * (EXP2_UNSCALE * EXP2_UNSCALE), m01 = g1.y * (EXP2_UNSCALE * EXP2_UNSCALE), m11 = g1.z * (EXP2_UNSCALE * EXP2_UNSCALE);          // sum q l l^T with the unscaled conic
            da = kk * 0.5f * m00; db = kk * m01; dc = kk * 0.5f * m11;
            if (p.dL_dconic) {          // the optional dL/dconic output of such a Gaussian, back from its sums: d = cov2D l, so sum q d d^T = cov2D (sum q l l^T) cov2D
                p.dL_dconic[4 * i] = -0.5f * (a_ * a_ * m00 + 2 * a_ * b_ * m01 + b_ * b_ * m11);
                p.dL_dconic[4 * i + 1] = -0.5f * (a_ * b_ * m00 + (a_ * c_ + b_ * b_) * m01 + b_ * c_ * m11);
                p.dL_dconic[4 * i + 2] = 0.f;
                p.dL_dconic[4 * i + 3] = -0.5f * (b_ * b_ * m00 + 2 * b_ * c_ * m01 + c_ * c_ * m11);
            }
        }
        gS[0] = (A[0][0] * A[0][0] * da + A[0][0] * A[1][0] * db + A[1][0] * A[1][0] * dc);
        gS[3] = (A[0][1] * A[0][1] * da + A[0][1] * A[1][1] * db + A[1][1] * A[1][1] * dc);
        gS[5] = (A[0][2] * A[0][2] * da + A[0][2] * A[1][2] * db + A[1][2] * A[1][2] * dc);
        gS[1] = 2 * A[0][0] * A[0][1] * da + (A[0][0] * A[1][1] + A[0][1] * A[1][0]) * db + 2 * A[1][0] * A[1][1] * dc;
        gS[2] = 2 * A[0][0] * A[0][2] * da + (A[0][0] * A[1][2] + A[0][2] * A[1][0]) * db + 2 * A[1][0] * A[1][2] * dc;
        gS[4] = 2 * A[0][2] * A[0][1] * da + (A[0][1] * A[1][2] + A[0][2] * A[1][1]) * db + 2 * A[1][1] * A[1][2] * dc;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) if (p.dL_dcov3D) p.dL_dcov3D[6 * i + k] = gS[k];
    float dA[2][3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float a0S = A[0][0] * S[r][0] + A[0][1] * S[r][1] + A[0][2] * S[r][2];
        const float a1S = A[1][0] * S[r][0] + A[1][1] * S[r][1] + A[1][2] * S[r][2];
        dA[0][r] = 2 * a0S * da + a1S * db;
        dA[1][r] = 2 * a1S * dc + a0S * db;
    }
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
                   vm[8] * dtx + vm[9] * dty + vm[10] * dtz};

    // ---- 2D mean -> 3D mean ----
    const float hw = pm[3] * mean[0] + pm[7] * mean[1] + pm[11] * mean[2] + pm[15];
    const float mw = 1.0f / (hw + 0.0000001f);
    const float mul1 = (pm[0] * mean[0] + pm[4] * mean[1] + pm[8] * mean[2] + pm[12]) * mw * mw;
    const float mul2 = (pm[1] * mean[0] + pm[5] * mean[1] + pm[9] * mean[2] + pm[13]) * mw * mw;
    gm[0] += (pm[0] * mw - pm[3] * mul1) * g2x + (pm[1] * mw - pm[3] * mul2) * g2y;
    gm[1] += (pm[4] * mw - pm[7] * mul1) * g2x + (pm[5] * mw - pm[7] * mul2) * g2y;
    gm[2] += (pm[8] * mw - pm[11] * mul1) * g2x + (pm[9] * mw - pm[11] * mul2) * g2y;

    // ---- SH backward ----
    if (p.shs) {
        const float dorig[3] = {mean[0] - cam.campos[0], mean[1] - cam.campos[1], mean[2] - cam.campos[2]};
        const float len = sqrtf(dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2]);
        const float x = dorig[0] / len, y = dorig[1] / len, z = dorig[2] / len;
        const float* shg = p.shs + (size_t)i * (SPLIT ? 1 : p.M) * 3;          // (split: the DC row; SHK reads the others from shs_rest)
        float* gsh = p.dL_dsh + (size_t)i * p.M * 3;
        float shv[FAST16 ? 48 : 1];
        if (FAST16) {
            if (SPLIT) {          // staged by the whole wave before this branch: the lane's DC (3 words) and rest (45 words) out of LDS, odd strides
                const float* s_w = reinterpret_cast<const float*>(s_t);
                const float* rr = s_w + 45 * threadIdx.x;
                const float* rd = s_w + 2880 + 3 * threadIdx.x;
                shv[0] = rd[0]; shv[1] = rd[1]; shv[2] = rd[2];
#pragma unroll
                for (int k = 0; k < 45; k++) shv[3 + k] = rr[k];
            } else {
                const float4* r4 = reinterpret_cast<const float4*>(shg);
#pragma unroll
                for (int v = 0; v < 12; v++) { const float4 q = r4[v]; shv[4 * v] = q.x; shv[4 * v + 1] = q.y; shv[4 * v + 2] = q.z; shv[4 * v + 3] = q.w; }
            }
        }
        const uint8_t cb = p.clamped[i];
        float g[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) g[ch] = gcol[ch] * (((cb >> ch) & 1) ? 0.f : 1.f);
        if (!WRITE_SH) { p.dL_dcolors[3 * i] = g[0]; p.dL_dcolors[3 * i + 1] = g[1]; p.dL_dcolors[3 * i + 2] = g[2]; }
        float B[16];
        const int D = p.D;
        const int nb = sh_basis(D, x, y, z, B);
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        if (!WRITE_SH) {
        } else if (FAST16) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k < nb) { gv[3 * k] = B[k] * g[0]; gv[3 * k + 1] = B[k] * g[1]; gv[3 * k + 2] = B[k] * g[2]; }
            }
        } else if (SPLIT) {
            float* gd = p.dL_dsh + (size_t)i * 3; float* gr = p.dL_dsh_rest + (size_t)i * (p.M - 1) * 3;
            gd[0] = B[0] * g[0]; gd[1] = B[0] * g[1]; gd[2] = B[0] * g[2];
            for (int k = 1; k < nb; k++) { gr[3 * (k - 1)] = B[k] * g[0]; gr[3 * (k - 1) + 1] = B[k] * g[1]; gr[3 * (k - 1) + 2] = B[k] * g[2]; }
            for (int k = 3 * (nb - 1); k < 3 * (p.M - 1); k++) gr[k] = 0.f;
        } else {
            for (int k = 0; k < nb; k++) {
                gsh[3 * k] = B[k] * g[0]; gsh[3 * k + 1] = B[k] * g[1]; gsh[3 * k + 2] = B[k] * g[2];
            }
            for (int k = 3 * nb; k < 3 * p.M; k++) gsh[k] = 0.f;      // coefficients above the active degree
        }
        float gd[3] = {0, 0, 0};
        if (D > 0) {
            float dx3[3], dy3[3], dz3[3];
#define SHK(k, ch) (FAST16 ? shv[3 * (k) + (ch)] : ((SPLIT && (k) > 0) ? p.shs_rest[((size_t)i * (p.M - 1) + ((k) - 1)) * 3 + (ch)] : shg[3 * (k) + (ch)]))
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                dx3[ch] = -bC1 * SHK(3, ch); dy3[ch] = -bC1 * SHK(1, ch); dz3[ch] = bC1 * SHK(2, ch);
                if (D > 1) {
                    dx3[ch] += bC2[0] * y * SHK(4, ch) + bC2[2] * 2.f * -x * SHK(6, ch) + bC2[3] * z * SHK(7, ch) + bC2[4] * 2.f * x * SHK(8, ch);
                    dy3[ch] += bC2[0] * x * SHK(4, ch) + bC2[1] * z * SHK(5, ch) + bC2[2] * 2.f * -y * SHK(6, ch) + bC2[4] * 2.f * -y * SHK(8, ch);
                    dz3[ch] += bC2[1] * y * SHK(5, ch) + bC2[2] * 2.f * 2.f * z * SHK(6, ch) + bC2[3] * x * SHK(7, ch);
                    if (D > 2) {
                        dx3[ch] += (bC3[0] * SHK(9, ch) * 3.f * 2.f * xy + bC3[1] * SHK(10, ch) * yz + bC3[2] * SHK(11, ch) * -2.f * xy +
                                    bC3[3] * SHK(12, ch) * -3.f * 2.f * xz + bC3[4] * SHK(13, ch) * (-3.f * xx + 4.f * zz - yy) +
                                    bC3[5] * SHK(14, ch) * 2.f * xz + bC3[6] * SHK(15, ch) * 3.f * (xx - yy));
                        dy3[ch] += (bC3[0] * SHK(9, ch) * 3.f * (xx - yy) + bC3[1] * SHK(10, ch) * xz + bC3[2] * SHK(11, ch) * (-3.f * yy + 4.f * zz - xx) +
                                    bC3[3] * SHK(12, ch) * -3.f * 2.f * yz + bC3[4] * SHK(13, ch) * -2.f * xy +
                                    bC3[5] * SHK(14, ch) * -2.f * yz + bC3[6] * SHK(15, ch) * -3.f * 2.f * xy);
                        dz3[ch] += (bC3[1] * SHK(10, ch) * xy + bC3[2] * SHK(11, ch) * 4.f * 2.f * yz + bC3[3] * SHK(12, ch) * 3.f * (2.f * zz - xx - yy) +
                                    bC3[4] * SHK(13, ch) * 4.f * 2.f * xz + bC3[5] * SHK(14, ch) * (xx - yy));
                    }
                }
            }
#undef SHK
            gd[0] = dx3[0] * g[0] + dx3[1] * g[1] + dx3[2] * g[2];
            gd[1] = dy3[0] * g[0] + dy3[1] * g[1] + dy3[2] * g[2];
            gd[2] = dz3[0] * g[0] + dz3[1] * g[1] + dz3[2] * g[2];
        }
        const float s2 = dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2];
        const float inv32 = 1.0f / sqrtf(s2 * s2 * s2);
        gm[0] += ((+s2 - dorig[0] * dorig[0]) * gd[0] - dorig[1] * dorig[0] * gd[1] - dorig[2] * dorig[0] * gd[2]) * inv32;
        gm[1] += (-dorig[0] * dorig[1] * gd[0] + (s2 - dorig[1] * dorig[1]) * gd[1] - dorig[2] * dorig[1] * gd[2]) * inv32;
        gm[2] += (-dorig[0] * dorig[2] * gd[0] - dorig[1] * dorig[2] * gd[1] + (s2 - dorig[2] * dorig[2]) * gd[2]) * inv32;
    }
    p.dL_dmean3D[3 * i] = gm[0] + pgm[0]; p.dL_dmean3D[3 * i + 1] = gm[1] + pgm[1]; p.dL_dmean3D[3 * i + 2] = gm[2] + pgm[2];

    // ---- cov3D -> scale / rotation ----
    if (p.scales) {
        const float r_ = p.rotations[4 * i], x = p.rotations[4 * i + 1], y = p.rotations[4 * i + 2], z = p.rotations[4 * i + 3];
        float R[3][3];
        R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r_ * z);      R[0][2] = 2.f * (x * z + r_ * y);
        R[1][0] = 2.f * (x * y + r_ * z);      R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r_ * x);
        R[2][0] = 2.f * (x * z - r_ * y);      R[2][1] = 2.f * (y * z + r_ * x);      R[2][2] = 1.f - 2.f * (x * x + y * y);
        const float sv[3] = {p.scale_modifier * p.scales[3 * i], p.scale_modifier * p.scales[3 * i + 1], p.scale_modifier * p.scales[3 * i + 2]};
        float Mm[3][3], dS[3][3], dM[3][3], G[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) Mm[r][cc] = sv[r] * R[cc][r];
        dS[0][0] = gS[0]; dS[1][1] = gS[3]; dS[2][2] = gS[5];
        dS[0][1] = dS[1][0] = 0.5f * gS[1]; dS[0][2] = dS[2][0] = 0.5f * gS[2]; dS[1][2] = dS[2][1] = 0.5f * gS[4];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) dM[r][cc] = 2.0f * (Mm[r][0] * dS[0][cc] + Mm[r][1] * dS[1][cc] + Mm[r][2] * dS[2][cc]);
#pragma unroll
        for (int r = 0; r < 3; r++) p.dL_dscale[3 * i + r] = R[0][r] * dM[r][0] + R[1][r] * dM[r][1] + R[2][r] * dM[r][2];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) G[r][cc] = sv[r] * dM[r][cc];
        p.dL_drot[4 * i + 0] = prot[0] + 2 * z * (G[0][1] - G[1][0]) + 2 * y * (G[2][0] - G[0][2]) + 2 * x * (G[1][2] - G[2][1]);
        p.dL_drot[4 * i + 1] = prot[1] + 2 * y * (G[1][0] + G[0][1]) + 2 * z * (G[2][0] + G[0][2]) + 2 * r_ * (G[1][2] - G[2][1]) - 4 * x * (G[2][2] + G[1][1]);
        p.dL_drot[4 * i + 2] = prot[2] + 2 * x * (G[1][0] + G[0][1]) + 2 * r_ * (G[2][0] - G[0][2]) + 2 * z * (G[1][2] + G[2][1]) - 4 * y * (G[2][2] + G[0][0]);
        p.dL_drot[4 * i + 3] = prot[3] + 2 * r_ * (G[0][1] - G[1][0]) + 2 * x * (G[2][0] + G[0][2]) + 2 * y * (G[1][2] + G[2][1]) - 4 * z * (G[1][1] + G[0][0]);
    }
    }   // vis

    if (FAST16 && WRITE_SH && SPLIT && p.shs) {
        // ... and out again the same way: every lane stores its 3 + 45 gradient words where the two arrays want them (odd strides: no conflicts), then
        // the block's two contiguous pieces of dL/dsh (720 + 48 float4) leave with coalesced 16-B writes
        const int lane = threadIdx.x;
        __syncthreads();          // (every lane has read its coefficient row)
        float* s_w = reinterpret_cast<float*>(s_t);
        s_w[2880 + 3 * lane] = gv[0]; s_w[2880 + 3 * lane + 1] = gv[1]; s_w[2880 + 3 * lane + 2] = gv[2];
#pragma unroll
        for (int k = 0; k < 45; k++) s_w[45 * lane + k] = gv[3 + k];
        __syncthreads();
        const int i0 = blockIdx.x * 64;
        const int nrows = min(64, p.P - i0);
        float4* dr = reinterpret_cast<float4*>(p.dL_dsh_rest + (size_t)i0 * 45);
        float4* dd = reinterpret_cast<float4*>(p.dL_dsh + (size_t)i0 * 3);
#pragma unroll
        for (int it = 0; it < 12; it++) {
            const int q = it * 64 + lane;
            const float4 f4 = s_t[q];
            const float f[4] = {f4.x, f4.y, f4.z, f4.w};
            if (q < 720) {
                if (4 * q + 3 < nrows * 45) dr[q] = f4;
                else { float* op = p.dL_dsh_rest + (size_t)i0 * 45; for (int k = 0; k < 4; k++) if (4 * q + k < nrows * 45) op[4 * q + k] = f[k]; }
            } else {
                const int qd = q - 720;
                if (4 * qd + 3 < nrows * 3) dd[qd] = f4;
                else { float* op = p.dL_dsh + (size_t)i0 * 3; for (int k = 0; k < 4; k++) if (4 * qd + k < nrows * 3) op[4 * qd + k] = f[k]; }
            }
        }
    } else if (FAST16 && WRITE_SH && p.shs) {
        const int lane = threadIdx.x;
#pragma unroll
        for (int v = 0; v < 12; v++) s_t[lane * 13 + v] = make_float4(gv[4 * v], gv[4 * v + 1], gv[4 * v + 2], gv[4 * v + 3]);
        __syncthreads();
        const int i0 = blockIdx.x * 64;
        const int nrows = min(64, p.P - i0);
        float4* dst = reinterpret_cast<float4*>(p.dL_dsh + (size_t)i0 * 48);
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int f = lane + 64 * k;          // quad index inside the block's contiguous 64 x 48 floats
            const int r = f / 12, c = f - 12 * r;
            if (r < nrows) dst[f] = s_t[r * 13 + c];
        }
    }
}

int launch_preprocess_backward(hipStream_t s, const ibgs_backward_args& a, const GeomState& g)
{
    PreBwdParams p;
    p.P = a.P; p.D = a.D; p.M = a.M;
    p.means3D = a.means3D; p.radii = a.radii; p.shs = a.shs; p.shs_rest = a.shs_rest; p.clamped = g.clamped;
    p.scales = a.scales; p.rotations = a.rotations; p.scale_modifier = a.scale_modifier;
    p.cov3D = a.cov3D_precomp ? a.cov3D_precomp : g.cov3D;
    p.gacc = a.grad_acc; p.rec = g.rec; p.clear_gacc = (a.flags & IBGS_FLAG_CLEAR_GRAD_ACC) ? 1 : 0;
    { const int ra = render_backward_ref_arith(a.flags); p.risky_rows = (ra & 4) ? 2 : ((ra & 2) ? 1 : 0); }          // RA_ASSOC / RA_LFORM (render_bwd.hip)
    p.dL_dmean2D = a.dL_dmean2D; p.dL_dmean2D_abs = a.dL_dmean2D_abs; p.dL_dconic = a.dL_dconic;
    p.dL_dopacity = a.dL_dopacity; p.dL_dcolors = a.dL_dcolors; p.dL_dall_map = a.dL_dall_map;
    p.dL_dmean3D = a.dL_dmean3D; p.dL_dcov3D = a.dL_dcov3D; p.dL_dsh = a.dL_dsh; p.dL_dsh_rest = a.dL_dsh_rest; p.dL_dscale = a.dL_dscale; p.dL_drot = a.dL_drot;
    p.plane_normal = a.plane_normal; p.plane_offset = a.plane_offset; p.plane_mode = a.plane_mode;
    p.dL_dplane_normal = a.dL_dplane_normal; p.dL_dplane_offset = a.dL_dplane_offset;
    const Cam cam = make_cam(a.viewmatrix, a.projmatrix, a.campos, a.bg, a.tanfovx, a.tanfovy, a.W, a.H);
    const bool factored = a.shs && (a.flags & IBGS_FLAG_SH_FACTORED);
    const bool split = a.shs && a.shs_rest;
    if (a.shs && a.M == 16) {
        if (factored && split) hipLaunchKernelGGL((preprocess_bwd_kernel<true, false, true>), dim3((a.P + 63) / 64), dim3(64), 0, s, p, cam);
        else if (factored) hipLaunchKernelGGL((preprocess_bwd_kernel<true, false>), dim3((a.P + 63) / 64), dim3(64), 0, s, p, cam);
        else if (split) hipLaunchKernelGGL((preprocess_bwd_kernel<true, true, true>), dim3((a.P + 63) / 64), dim3(64), 0, s, p, cam);
        else hipLaunchKernelGGL((preprocess_bwd_kernel<true, true>), dim3((a.P + 63) / 64), dim3(64), 0, s, p, cam);
    } else {
        if (factored && split) hipLaunchKernelGGL((preprocess_bwd_kernel<false, false, true>), dim3((a.P + 255) / 256), dim3(256), 0, s, p, cam);
        else if (factored) hipLaunchKernelGGL((preprocess_bwd_kernel<false, false>), dim3((a.P + 255) / 256), dim3(256), 0, s, p, cam);
        else if (split) hipLaunchKernelGGL((preprocess_bwd_kernel<false, true, true>), dim3((a.P + 255) / 256), dim3(256), 0, s, p, cam);
        else hipLaunchKernelGGL((preprocess_bwd_kernel<false, true>), dim3((a.P + 255) / 256), dim3(256), 0, s, p, cam);
    }
    IBGS_HIP(hipGetLastError());
    return 0;
}

// dL/dsh summed over the views of a view-parallel step, rebuilt from the exchanged factors:
// dL_dsh[i][k][c] = sum_v basis_k(normalise(mean_i - campos_v)) * dcolor[v][i][c]   (k < (D+1)^2, zero above).
// One thread per Gaussian; views in index order on every rank, so all ranks hold bit-identical sums.
// M == 16: the 192-B rows leave through the same LDS transpose as preprocess_bwd_kernel<true, true>.
template <bool FAST16>
__global__ void __launch_bounds__(FAST16 ? 64 : 256) sh_grad_from_views_kernel(int P, int D, int M, int n_views, size_t view_stride, const float* __restrict__ means3D,
                                                                                const float* __restrict__ camposes, const float* __restrict__ dcolor,
                                                                                float* __restrict__ dL_dsh)
{
    __shared__ float4 s_t[FAST16 ? 64 * 13 : 1];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float acc[48];
#pragma unroll
    for (int k = 0; k < 48; k++) acc[k] = 0.f;
    if (i < P) {
        const float mx = means3D[3 * i], my = means3D[3 * i + 1], mz = means3D[3 * i + 2];
        for (int v = 0; v < n_views; v++) {
            const float* g = dcolor + (size_t)v * view_stride + (size_t)i * 3;
            const float g0 = g[0], g1 = g[1], g2 = g[2];
            const float dx = mx - camposes[3 * v], dy = my - camposes[3 * v + 1], dz = mz - camposes[3 * v + 2];
            const float len = sqrtf(dx * dx + dy * dy + dz * dz);
            float B[16];
            const int nb = sh_basis(D, dx / len, dy / len, dz / len, B);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k < nb) { acc[3 * k] += B[k] * g0; acc[3 * k + 1] += B[k] * g1; acc[3 * k + 2] += B[k] * g2; }
            }
        }
    }
    if (FAST16) {
        const int lane = threadIdx.x;
#pragma unroll
        for (int v = 0; v < 12; v++) s_t[lane * 13 + v] = make_float4(acc[4 * v], acc[4 * v + 1], acc[4 * v + 2], acc[4 * v + 3]);
        __syncthreads();
        const int i0 = blockIdx.x * 64;
        const int nrows = min(64, P - i0);
        float4* dst = reinterpret_cast<float4*>(dL_dsh + (size_t)i0 * 48);
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int f = lane + 64 * k;
            const int r = f / 12, c = f - 12 * r;
            if (r < nrows) dst[f] = s_t[r * 13 + c];
        }
    } else if (i < P) {
        float* gsh = dL_dsh + (size_t)i * M * 3;
        const int nact = min(M, 16);
#pragma unroll
        for (int k = 0; k < 16; k++) if (k < nact) { gsh[3 * k] = acc[3 * k]; gsh[3 * k + 1] = acc[3 * k + 1]; gsh[3 * k + 2] = acc[3 * k + 2]; }          // (static indices: a run-time one sends acc[] to scratch memory)
        for (int k = 3 * nact; k < 3 * M; k++) gsh[k] = 0.f;
    }
}

int launch_sh_grad_from_views(hipStream_t s, int P, int D, int M, int n_views, const float* means3D, const float* camposes,
                              const float* dcolor, size_t view_stride, float* dL_dsh)
{
    if (M == 16) hipLaunchKernelGGL(sh_grad_from_views_kernel<true>, dim3((P + 63) / 64), dim3(64), 0, s, P, D, M, n_views, view_stride, means3D, camposes, dcolor, dL_dsh);
    else hipLaunchKernelGGL(sh_grad_from_views_kernel<false>, dim3((P + 255) / 256), dim3(256), 0, s, P, D, M, n_views, view_stride, means3D, camposes, dcolor, dL_dsh);
    IBGS_HIP(hipGetLastError());
    return 0;
}

// ---- the optimiser step of the SH coefficients straight from the factors (round 6) --------------------------------------------------------------------
// A single-GPU trainer with FusedAdam (optim.py) pays for dL/dsh twice: preprocess_bwd writes the dense (P, 16, 3) gradient -- 192 B per Gaussian, two thirds of
// everything that kernel writes -- and the Adam kernel reads it back, although for one view it is the outer product basis(dir) x dL/dRGB of 3 + 3 floats.  Here the
// gradient of a workgroup's 64 Gaussians is rebuilt in LDS (the loop of sh_grad_from_views_kernel above: the same products, summed over the views in index order)
// and consumed on the spot by the update of the coefficient tensors -- `f_dc` (P, 1, 3) and `f_rest` (P, M - 1, 3) with their own learning rates, or one combined
// (P, M, 3) tensor: tensor t holds coefficients k0[t] .. k0[t] + K[t] - 1 of every Gaussian.  The update is adam_math.h's, so parameters and moments are bit-identical
// to ibgs_sh_grad_from_views + ibgs_adam_step (tests/test_gpu_adam.py).  HBM-bound like adam_kernel: 6 floats of traffic per coefficient instead of 7 + the write.
struct AdamShParams {
    int P, D, n_views, n; size_t view_stride;
    const float* means3D; const float* camposes; const float* dcolor;
    AdamDev t[2]; int k0[2], K3[2]; uint32_t inv24[2];          // K3 = 3 K; inv24 = ceil(2^24 / K3): e / K3 == (e * inv24) >> 24 for e < 64 K3 (K <= 16)
};
constexpr int ASH_ROW = 49;          // LDS words per Gaussian: 48 gradient values + 1 of padding
__global__ void __launch_bounds__(256) adam_sh_kernel(AdamShParams q)
{
    __shared__ float s_g[64 * ASH_ROW];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * 64;
    if (tid < 64) {
        const int i = i0 + tid;
        float acc[48];
#pragma unroll
        for (int k = 0; k < 48; k++) acc[k] = 0.f;
        if (i < q.P) {
            const float mx = q.means3D[3 * i], my = q.means3D[3 * i + 1], mz = q.means3D[3 * i + 2];
            for (int v = 0; v < q.n_views; v++) {
                const float* g = q.dcolor + (size_t)v * q.view_stride + (size_t)i * 3;
                const float g0 = g[0], g1 = g[1], g2 = g[2];
                const float dx = mx - q.camposes[3 * v], dy = my - q.camposes[3 * v + 1], dz = mz - q.camposes[3 * v + 2];
                const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                float B[16];
                const int nb = sh_basis(q.D, dx / len, dy / len, dz / len, B);
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (k < nb) { acc[3 * k] += B[k] * g0; acc[3 * k + 1] += B[k] * g1; acc[3 * k + 2] += B[k] * g2; }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 48; k++) s_g[tid * ASH_ROW + k] = acc[k];
    }
    __syncthreads();
    const int nrows = min(64, q.P - i0);
    for (int t = 0; t < q.n; t++) {          // uniform
        const AdamDev d = q.t[t];
        const int K3 = q.K3[t], off = 3 * q.k0[t];
        const uint32_t inv = q.inv24[t];
        const size_t base = (size_t)i0 * K3;          // (a multiple of 64 floats: the slab starts 16-byte aligned whenever the tensor does)
        const int count = nrows * K3;
        auto grad_of = [&](int e) { const int r = (int)(((uint32_t)e * inv) >> 24); return s_g[r * ASH_ROW + off + (e - r * K3)]; };
        const bool vec_ok = ((reinterpret_cast<uintptr_t>(d.param) | reinterpret_cast<uintptr_t>(d.exp_avg) | reinterpret_cast<uintptr_t>(d.exp_avg_sq)) & 15u) == 0;
        const int nvec = vec_ok ? (count & ~3) : 0;
        for (int e = tid * 4; e < nvec; e += 256 * 4) {
            float4 p = *reinterpret_cast<float4*>(d.param + base + e);
            float4 m = *reinterpret_cast<float4*>(d.exp_avg + base + e), v = *reinterpret_cast<float4*>(d.exp_avg_sq + base + e);
            adam_one(p.x, grad_of(e), m.x, v.x, d);
            adam_one(p.y, grad_of(e + 1), m.y, v.y, d);
            adam_one(p.z, grad_of(e + 2), m.z, v.z, d);
            adam_one(p.w, grad_of(e + 3), m.w, v.w, d);
            *reinterpret_cast<float4*>(d.param + base + e) = p;
            *reinterpret_cast<float4*>(d.exp_avg + base + e) = m; *reinterpret_cast<float4*>(d.exp_avg_sq + base + e) = v;
        }
        for (int e = nvec + tid; e < count; e += 256) {
            float p = d.param[base + e], m = d.exp_avg[base + e], v = d.exp_avg_sq[base + e];
            adam_one(p, grad_of(e), m, v, d);
            d.param[base + e] = p; d.exp_avg[base + e] = m; d.exp_avg_sq[base + e] = v;
        }
    }
}

}  // namespace ibgs

extern "C" int32_t ibgs_adam_step_sh(void* stream, int32_t P, int32_t D, int32_t n_views, const float* means3D, const float* camposes, const float* dcolor,
                                     int64_t view_stride, int32_t n_tensors, const ibgs_adam_tensor* tensors, const int32_t* first_coeff, const int32_t* n_coeff)
{
    using namespace ibgs;
    if (P <= 0 || n_tensors <= 0) return 0;
    if (n_tensors > 2 || !tensors || !first_coeff || !n_coeff) { set_error("ibgs_adam_step_sh: 1 or 2 coefficient tensors"); return -IBGS_ERR_INVALID; }
    if (D < 0 || D > 3 || n_views < 1 || !means3D || !camposes || !dcolor) { set_error("ibgs_adam_step_sh: D in 0..3, n_views >= 1, non-null factors"); return -IBGS_ERR_INVALID; }
    AdamShParams q;
    q.P = P; q.D = D; q.n_views = n_views; q.n = 0; q.view_stride = view_stride > 0 ? (size_t)view_stride : (size_t)P * 3;
    q.means3D = means3D; q.camposes = camposes; q.dcolor = dcolor;
    for (int t = 0; t < n_tensors; t++) {
        const ibgs_adam_tensor& d = tensors[t];
        const int k0 = first_coeff[t], K = n_coeff[t];
        if (K <= 0) continue;
        if (k0 < 0 || k0 + K > 16) { set_error("ibgs_adam_step_sh: tensor %d holds coefficients %d .. %d (of 16)", t, k0, k0 + K - 1); return -IBGS_ERR_INVALID; }
        if (!d.param || !d.exp_avg || !d.exp_avg_sq || d.numel != (int64_t)P * K * 3) { set_error("ibgs_adam_step_sh: tensor %d must hold P x %d x 3 floats", t, K); return -IBGS_ERR_INVALID; }
        if (!(d.bias_correction1 > 0.f) || !(d.bias_correction2 > 0.f)) { set_error("ibgs_adam_step_sh: bias corrections must be positive"); return -IBGS_ERR_INVALID; }
        q.t[q.n] = adam_dev_from(d); q.t[q.n].grad = nullptr;
        q.k0[q.n] = k0; q.K3[q.n] = 3 * K; q.inv24[q.n] = (uint32_t)(((1u << 24) + 3u * (uint32_t)K - 1u) / (3u * (uint32_t)K));
        q.n++;
    }
    if (q.n == 0) return 0;
    hipLaunchKernelGGL(adam_sh_kernel, dim3((unsigned)((P + 63) / 64)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), q);
    IBGS_HIP(hipGetLastError());
    return 0;
}
