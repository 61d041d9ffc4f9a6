// B1 / B2: back-to-front re-traversal of every tile list, gradients w.r.t. the per-Gaussian render
// record (2D mean, conic, opacity, colour, plane parameters).
//
// Behaviour: DPR/cuda_rasterizer/backward.cu:496-807 (renderCUDA) + bilinearInterpolateBackward
// (backward.cu:55-109).  The reference issues up to 16 global atomicAdd per (pixel, Gaussian) pair
// into five separate arrays.  On MI355X global float atomics run at ~1.3 TB/s chip-wide only when
// they arrive as contiguous 64-byte requests (MI355X_MICROARCH.md "Global float atomics"), so here:
//
//   * one wave owns a 16x16 tile (PPL = 4) or an 8x8 quadrant (PPL = 1, geo variant); a lane first
//     sums its own pixels' contributions in registers,
//   * the 64 lanes are reduced with DPP row operations + readlane (no LDS traffic, no LDS atomics),
//   * the wave then issues ONE atomic instruction per Gaussian whose lanes 0..15 add the 16 floats of
//     that Gaussian's 64-byte accumulation row (grad_acc[P][16]) -- one 64-byte memory-side request
//     per (Gaussian, tile) instead of 11-16 scattered dword atomics per (Gaussian, pixel);
//   * Gaussians that no pixel of the wave uses (ballot == 0, or behind every pixel's last
//     contributor) cost a handful of VALU instructions and no memory traffic.
//
// preprocess_bwd.hip later converts the accumulation rows into the reference's output layout.
//
// Deviation (documented in DESIGN.md): alpha is recomputed with the same fast exp as the forward
// (the reference uses __expf forward / exp backward), so T/(1-alpha) retraces the
// forward transmittance exactly.
#include "common.h"

namespace ibgs {

struct BwdParams {
    const uint32_t* ranges; const uint32_t* point_list; const float4* rec;
    Cam cam;
    int ntiles;
    int n_src; int tex_quant;
    const float* ref_to_src; const float4* src_rgba;
    const float* final_T; const uint32_t* n_contrib; const float* sum_w; const uint32_t* low_high;
    const int32_t* valid_idx; const float* valid_w;
    const float* depth_pixels; const float* warped_pixels;
    const float* dL_dcolor; const float* dL_dnormal; const float* dL_ddepth; const float* dL_dwarped;
    float* gacc;
};

__device__ __forceinline__ float quant8b(float a, int quant) { return quant ? floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f) : a; }

__device__ __forceinline__ float4 tex_rgba_b(const float4* __restrict__ img, int W, int H, float x, float y, int quant)
{
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

// Sum over the 64 lanes of a wave; result valid in every lane.
// DPP within rows of 16 (quad_perm / row_half_mirror / row_mirror), then the four row totals are
// combined through readlane (SGPR broadcast).
__device__ __forceinline__ float wave_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false)); // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false)); // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, WAVE));
    return v;
}

__device__ __forceinline__ int xcd_band_map_b(int b, int n)
{
    const int per = (n + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}

template <bool GEO, int PPL>
__global__ void __launch_bounds__(64) render_bwd_kernel(BwdParams p)
{
    constexpr int NQ = GEO ? 4 : 3;
    __shared__ float4 s_rec[NQ][WAVE];
    __shared__ uint32_t s_id[WAVE];

    const int lane = threadIdx.x;
    const int nitems = p.ntiles * (PPL == 4 ? 1 : 4);
    const int item = xcd_band_map_b(blockIdx.x, nitems);
    if (item >= nitems) return;
    const int tile = (PPL == 4) ? item : (item >> 2);
    const int quad0 = (PPL == 4) ? 0 : (item & 3);
    const int W = p.cam.W, H = p.cam.H;
    const int tx0 = (tile % p.cam.gx) * TILE, ty0 = (tile / p.cam.gx) * TILE;
    const size_t HW = (size_t)W * H;
    const float fx = p.cam.fx, fy = p.cam.fy;
    const float cx = (float)(W * 0.5f), cy = (float)(H * 0.5f);
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);

    float pxf[PPL], pyf[PPL];
    size_t pixid[PPL];
    bool inside[PPL];
    float T[PPL], T_final[PPL], last_alpha[PPL], last_color[PPL][3], accum_rec[PPL][3], g_pix[PPL][3], bg_dot[PPL];
    uint32_t ncontrib[PPL];
    // geo
    float last_n[PPL][3], accum_n[PPL][3], g_n[PPL][3], g_d[PPL], rayx[PPL], rayy[PPL];
    uint32_t min_med[PPL], max_med[PPL];

    uint32_t nmax = 0;
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        const int qq = quad0 + q;
        const int px = tx0 + (qq & 1) * 8 + (lane & 7), py = ty0 + (qq >> 1) * 8 + (lane >> 3);
        pxf[q] = (float)px; pyf[q] = (float)py;
        inside[q] = px < W && py < H;
        pixid[q] = (size_t)py * W + px;
        T_final[q] = inside[q] ? p.final_T[pixid[q]] : 0.f;
        T[q] = T_final[q];
        ncontrib[q] = inside[q] ? p.n_contrib[pixid[q]] : 0u;
        nmax = max(nmax, ncontrib[q]);
        last_alpha[q] = 0.f; bg_dot[q] = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            last_color[q][ch] = 0.f; accum_rec[q][ch] = 0.f;
            g_pix[q][ch] = (inside[q] && p.dL_dcolor) ? p.dL_dcolor[ch * HW + pixid[q]] : 0.f;
            bg_dot[q] += p.cam.bg[ch] * g_pix[q][ch];
        }
        if (GEO) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                last_n[q][ch] = 0.f; accum_n[q][ch] = 0.f;
                g_n[q][ch] = (inside[q] && p.dL_dnormal) ? p.dL_dnormal[ch * HW + pixid[q]] : 0.f;
            }
            g_d[q] = (inside[q] && p.dL_ddepth) ? p.dL_ddepth[pixid[q]] : 0.f;
            min_med[q] = inside[q] ? p.low_high[2 * pixid[q]] : 0u;
            max_med[q] = inside[q] ? p.low_high[2 * pixid[q] + 1] : 0u;
            // dbl: backward.cu:545 evaluates (pix - W*0.5)/fx in double
            rayx[q] = (float)(((double)pxf[q] - W * 0.5) / (double)fx);
            rayy[q] = (float)(((double)pyf[q] - H * 0.5) / (double)fy);
        }
    }
    nmax = wave_max_u32(nmax);
    const uint32_t r0 = p.ranges[2 * tile], r1 = p.ranges[2 * tile + 1];
    const int n = (int)(r1 - r0);
    int top = min((int)nmax, n);           // entries >= top contribute to no pixel of this wave

    while (top > 0) {
        const int count = min(WAVE, top);
        {   // stage in processing order: slot l holds entry top-1-l
            if (lane < count) {
                const uint32_t id = p.point_list[r0 + (uint32_t)(top - 1 - lane)];
                const float4* r = p.rec + (size_t)id * 4;
                s_id[lane] = id;
                s_rec[0][lane] = r[0]; s_rec[1][lane] = r[1]; s_rec[2][lane] = r[2];
                if (GEO) s_rec[3][lane] = r[3];
            }
        }
        __syncthreads();
        for (int j = 0; j < count; j++) {
            const uint32_t k = (uint32_t)(top - 1 - j);          // 0-based position in the tile list
            const float4 q0 = s_rec[0][j], q1 = s_rec[1][j], q2 = s_rec[2][j];
            float4 q3 = q2;
            if (GEO) q3 = s_rec[3][j];
            float s_mx = 0.f, s_my = 0.f, s_ax = 0.f, s_ay = 0.f, s_ca = 0.f, s_cb = 0.f, s_cc = 0.f, s_op = 0.f;
            float s_r = 0.f, s_g = 0.f, s_b = 0.f, s_nx = 0.f, s_ny = 0.f, s_nz = 0.f, s_dist = 0.f;
            bool any = false;
#pragma unroll
            for (int q = 0; q < PPL; q++) {
                const float dx = q0.x - pxf[q], dy = q0.y - pyf[q];
                const float power = -0.5f * (q1.x * dx * dx + q1.z * dy * dy) - q1.y * dx * dy;
                const float G = __expf(power);
                const float alpha = fminf(0.99f, q0.z * G);
                const bool ok = (k < ncontrib[q]) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                if (__ballot(ok) != 0ull) {
                    any = true;
                    if (ok) {
                        T[q] = T[q] / (1.f - alpha);
                        const float w = alpha * T[q];
                        float dL_dalpha = 0.0f;
                        const float col[3] = {q2.x, q2.y, q2.z};
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            accum_rec[q][ch] = last_alpha[q] * last_color[q][ch] + (1.f - last_alpha[q]) * accum_rec[q][ch];
                            last_color[q][ch] = col[ch];
                            dL_dalpha += (col[ch] - accum_rec[q][ch]) * g_pix[q][ch];
                        }
                        s_r += w * g_pix[q][0]; s_g += w * g_pix[q][1]; s_b += w * g_pix[q][2];
                        if (GEO) {
                            const float nrm[3] = {q3.x, q3.y, q3.z};
                            float gm0 = 0.f, gm1 = 0.f, gm2 = 0.f, gm4 = 0.f;
#pragma unroll
                            for (int ch = 0; ch < 3; ch++) {
                                accum_n[q][ch] = last_alpha[q] * last_n[q][ch] + (1.f - last_alpha[q]) * accum_n[q][ch];
                                last_n[q][ch] = nrm[ch];
                                dL_dalpha += (nrm[ch] - accum_n[q][ch]) * g_n[q][ch];
                            }
                            gm0 += w * g_n[q][0]; gm1 += w * g_n[q][1]; gm2 += w * g_n[q][2];
                            // unsigned comparison: min_med == 0 disables the branch
                            if ((k >= (uint32_t)((int)min_med[q] - 1)) && (k <= (uint32_t)((int)max_med[q] - 1))) {
                                const float dist = q1.w;
                                const float dotn = nrm[0] * rayx[q] + nrm[1] * rayy[q] + nrm[2];
                                const float tmp = (float)((double)dotn + 1.0e-8);                 // dbl, backward.cu:697
                                const float tmp2 = dist / (tmp * tmp);
                                const float dep = (float)(-(double)dist / ((double)dotn + 1.0e-8));  // dbl, backward.cu:699
                                if (dep > 0.0f) {
                                    const float X = (pxf[q] - cx) * dep / fx, Y = (pyf[q] - cy) * dep / fy, Z = dep;
                                    const float sumw = p.sum_w[pixid[q]];
                                    float gdep = g_d[q] * w / sumw;
                                    dL_dalpha += g_d[q] * (dep - p.depth_pixels[pixid[q]]) / sumw;
                                    for (int m = 0; m < IBGS_MAX_SRC; m++) {
                                        const int si = p.valid_idx[(size_t)m * HW + pixid[q]];
                                        if (si == -1) break;
                                        const float* r = p.ref_to_src + 16 * si;
                                        const float tx = r[0] * X + r[1] * Y + r[2] * Z + r[3];
                                        const float ty = r[4] * X + r[5] * Y + r[6] * Z + r[7];
                                        const float tz = r[8] * X + r[9] * Y + r[10] * Z + r[11];
                                        const float u = (tx * fx / tz) + cx, v = (ty * fy / tz) + cy;
                                        if (u >= 0 && u <= W - 1 && v >= 0 && v <= H - 1) {
                                            const float4* img = p.src_rgba + (size_t)si * HW;
                                            const float4 c4 = tex_rgba_b(img, W, H, u + 0.5f, v + 0.5f, p.tex_quant);
                                            const float cc[3] = {c4.x, c4.y, c4.z};
                                            const float sw = p.valid_w[(size_t)m * HW + pixid[q]];
                                            float gc[3];
#pragma unroll
                                            for (int ch = 0; ch < 3; ch++) {
                                                const float gw = p.dL_dwarped ? p.dL_dwarped[((size_t)m * 3 + ch) * HW + pixid[q]] : 0.f;
                                                gc[ch] = gw * w / sw;
                                                dL_dalpha += gw * (cc[ch] - p.warped_pixels[((size_t)m * 3 + ch) * HW + pixid[q]]) / sw;
                                            }
                                            const float Av = (pxf[q] - cx) / fx, Bv = (pyf[q] - cy) / fy;
                                            const float U = r[0] * Av + r[1] * Bv + r[2];
                                            const float V = r[4] * Av + r[5] * Bv + r[6];
                                            const float Wc = r[8] * Av + r[9] * Bv + r[10];
                                            const float den = (Wc * dep + r[11]);
                                            const float dpx = fx * (U * r[11] - Wc * r[3]) / (den * den);
                                            const float dpy = fy * (V * r[11] - Wc * r[7]) / (den * den);
                                            // four linear-filtered fetches at integer coordinates
                                            const float uu = u + 0.5f, vv = v + 0.5f;
                                            const int u0 = (int)floorf(uu), v0 = (int)floorf(vv);
                                            const float fu = uu - (float)u0, fv = vv - (float)v0, fu1 = 1.0f - fu, fv1 = 1.0f - fv;
                                            const float4 I00 = tex_rgba_b(img, W, H, (float)u0, (float)v0, p.tex_quant);
                                            const float4 I01 = tex_rgba_b(img, W, H, (float)(u0 + 1), (float)v0, p.tex_quant);
                                            const float4 I10 = tex_rgba_b(img, W, H, (float)u0, (float)(v0 + 1), p.tex_quant);
                                            const float4 I11 = tex_rgba_b(img, W, H, (float)(u0 + 1), (float)(v0 + 1), p.tex_quant);
                                            const float dIu0 = -fv1 * I00.x + fv1 * I01.x - fv * I10.x + fv * I11.x;
                                            const float dIu1 = -fv1 * I00.y + fv1 * I01.y - fv * I10.y + fv * I11.y;
                                            const float dIu2 = -fv1 * I00.z + fv1 * I01.z - fv * I10.z + fv * I11.z;
                                            const float dIv0 = -fu1 * I00.x - fu * I01.x + fu1 * I10.x + fu * I11.x;
                                            const float dIv1 = -fu1 * I00.y - fu * I01.y + fu1 * I10.y + fu * I11.y;
                                            const float dIv2 = -fu1 * I00.z - fu * I01.z + fu1 * I10.z + fu * I11.z;
                                            const float du = gc[0] * dIu0 + gc[1] * dIu1 + gc[2] * dIu2;
                                            const float dv = gc[0] * dIv0 + gc[1] * dIv1 + gc[2] * dIv2;
                                            gdep += du * dpx + dv * dpy;
                                            // emitted inside the per-source in-bounds branch
                                            gm4 += (-gdep / tmp);
                                            gm0 += gdep * tmp2 * rayx[q];
                                            gm1 += gdep * tmp2 * rayy[q];
                                            gm2 += gdep * tmp2;
                                        }
                                    }
                                }
                            }
                            s_nx += gm0; s_ny += gm1; s_nz += gm2; s_dist += gm4;
                        }
                        dL_dalpha *= T[q];
                        last_alpha[q] = alpha;
                        dL_dalpha += (-T_final[q] / (1.f - alpha)) * bg_dot[q];
                        const float dL_dG = q0.z * dL_dalpha;
                        const float gdx = G * dx, gdy = G * dy;
                        const float dG_ddelx = -gdx * q1.x - gdy * q1.y;
                        const float dG_ddely = -gdy * q1.z - gdx * q1.y;
                        const float mx = dL_dG * dG_ddelx * ddelx_dx, my = dL_dG * dG_ddely * ddely_dy;
                        s_mx += mx; s_my += my; s_ax += fabsf(mx); s_ay += fabsf(my);
                        s_ca += -0.5f * gdx * dx * dL_dG;
                        s_cb += -0.5f * gdx * dy * dL_dG;
                        s_cc += -0.5f * gdy * dy * dL_dG;
                        s_op += G * dL_dalpha;
                    }
                }
            }
            if (any) {   // wave-uniform
                const float t_mx = wave_sum(s_mx), t_my = wave_sum(s_my), t_ax = wave_sum(s_ax), t_ay = wave_sum(s_ay);
                const float t_ca = wave_sum(s_ca), t_cb = wave_sum(s_cb), t_cc = wave_sum(s_cc), t_op = wave_sum(s_op);
                const float t_r = wave_sum(s_r), t_g = wave_sum(s_g), t_b = wave_sum(s_b);
                float v = 0.f;
                v = (lane == G_MX) ? t_mx : v; v = (lane == G_MY) ? t_my : v;
                v = (lane == G_AX) ? t_ax : v; v = (lane == G_AY) ? t_ay : v;
                v = (lane == G_CA) ? t_ca : v; v = (lane == G_CB) ? t_cb : v; v = (lane == G_CC) ? t_cc : v;
                v = (lane == G_OP) ? t_op : v;
                v = (lane == G_R) ? t_r : v; v = (lane == G_G) ? t_g : v; v = (lane == G_B) ? t_b : v;
                if (GEO) {
                    const float t_nx = wave_sum(s_nx), t_ny = wave_sum(s_ny), t_nz = wave_sum(s_nz), t_d = wave_sum(s_dist);
                    v = (lane == G_NX) ? t_nx : v; v = (lane == G_NY) ? t_ny : v; v = (lane == G_NZ) ? t_nz : v;
                    v = (lane == G_DIST) ? t_d : v;
                }
                const uint32_t id = s_id[j];
                if (lane < (GEO ? 15 : 11)) atomicAdd(p.gacc + (size_t)id * GACC_FLOATS + lane, v);
            }
        }
        __syncthreads();
        top -= count;
    }
}

int launch_render_backward(hipStream_t s, const ibgs_backward_args& a, const GeomState& g, const BinState& b,
                           const ImgState& im, const float4* src_rgba)
{
    BwdParams p;
    p.ranges = im.ranges; p.point_list = b.point_list; p.rec = reinterpret_cast<const float4*>(g.rec);
    p.cam = make_cam(a.viewmatrix, a.projmatrix, a.campos, a.bg, a.tanfovx, a.tanfovy, a.W, a.H);
    p.ntiles = p.cam.gx * p.cam.gy;
    p.n_src = a.n_src; p.tex_quant = (a.flags & IBGS_FLAG_TEX_QUANT) ? 1 : 0;
    p.ref_to_src = a.ref_to_src; p.src_rgba = src_rgba;
    p.final_T = im.final_T; p.n_contrib = im.n_contrib; p.sum_w = im.sum_w; p.low_high = im.low_high;
    p.valid_idx = im.valid_idx; p.valid_w = im.valid_w;
    p.depth_pixels = a.out_depth; p.warped_pixels = a.out_warped;
    p.dL_dcolor = a.dL_dcolor; p.dL_dnormal = a.dL_dnormal; p.dL_ddepth = a.dL_ddepth; p.dL_dwarped = a.dL_dwarped;
    p.gacc = a.grad_acc;
    const int nt = p.ntiles;
    if (a.render_geo) {
        const int grid = ((nt * 4 + 7) / 8) * 8;
        hipLaunchKernelGGL((render_bwd_kernel<true, 1>), dim3(grid), dim3(64), 0, s, p);
    } else {
        const int grid = ((nt + 7) / 8) * 8;
        hipLaunchKernelGGL((render_bwd_kernel<false, 4>), dim3(grid), dim3(64), 0, s, p);
    }
    IBGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace ibgs
