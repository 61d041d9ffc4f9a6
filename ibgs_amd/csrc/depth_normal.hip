// Row G(vii) of SURVEY.md 8(a): the depth -> normal map of the reference's render glue, forward and backward, one kernel each.
//
// Behaviour: utils/graphics_utils.py:17-83 (`ndc_2_cam`, `depth2point_cam`, `depth_pcd2normal`, `normal_from_depth_image`, offset = None) through
// `render_normal` (gaussian_renderer/__init__.py:16-26, scale = 1) and the normalisation `render()` applies to its result (:338-342):
//
//     xs = (u / (W - 1)) * (W - 1),  ys likewise                      (the reference's ndc round trip, in its operation order)
//     P(u, v) = [xs d, ys d, d] @ inv(K^T)   = (xs d / fx - d cx / fx,  ys d / fy - d cy / fy,  d)
//     c = (P(u+1, v) - P(u-1, v)) x (P(u, v-1) - P(u, v+1))           interior pixels; the one-pixel border is zero
//     n = c / max(|c|, 1e-12)                                           F.normalize
//     out = n / (|n| + 1e-8)                                            render()
//
// In torch that is ~25 kernels per call (two aranges, a meshgrid, a stack, a (HW x 3) @ (3 x 3) product through rocBLAS -- 128 us for a 1080p
// image on MI355X, profiles/r05_train_iter_kernels.txt --, an LU inversion, slices, cross, norm, pad, permute, norm, div) and as many again in the
// backward: ~0.3 ms of a 3.1 ms trainer iteration.  Here: one pixel-parallel kernel each way, bound by its HBM traffic (forward 4 + 12 B per pixel,
// backward 4 + 12 + 4 B; the 3 x 3 / 13-point stencils come out of L1 / L2; the backward shares its per-pixel edge gradients through LDS: see its kernel).
//
// The backward is the exact derivative of the expression above (both normalisations, the clamp branch of F.normalize included), written as a GATHER:
// the depth of pixel q enters the normals of its four neighbours p (as their right / left / top / bottom point), so q takes dL/d(P_r - P_l) and
// dL/d(P_t - P_b) of each of them (computed once per pixel of a tile and its halo, parked in LDS) -- no atomics, no buffer in memory, bit-reproducible.
#include "common.h"

namespace ibgs {

struct DnCam { float ifx, ify, mcx, mcy; float wm1, hm1; int W, H; };          // 1/fx, 1/fy, -cx/fx, -cy/fy (inv(K^T) of an upper-triangular K without skew)

__device__ __forceinline__ void dn_ray(const DnCam& c, int u, int v, float& rx, float& ry)
{   // dP/dd of pixel (u, v): P = d * (rx, ry, 1)
#pragma clang fp contract(off)
    const float xs = ((float)u / c.wm1) * c.wm1, ys = ((float)v / c.hm1) * c.hm1;          // the reference's arange / (W - 1) * (W - 1)
    rx = xs * c.ifx + c.mcx; ry = ys * c.ify + c.mcy;
}

struct Vec3 { float x, y, z; };
__device__ __forceinline__ Vec3 dn_point(const DnCam& c, int u, int v, float d)
{   // [xs d, ys d, d] @ inv(K^T), term by term in the reference's order (no contraction: the normals are differences of neighbouring points)
#pragma clang fp contract(off)
    const float xs = ((float)u / c.wm1) * c.wm1, ys = ((float)v / c.hm1) * c.hm1;
    return {(xs * d) * c.ifx + d * c.mcx, (ys * d) * c.ify + d * c.mcy, d};
}
__device__ __forceinline__ Vec3 cross3(const Vec3& a, const Vec3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// a = P_right - P_left, b = P_top - P_bottom of interior pixel (u, v)
__device__ __forceinline__ void dn_edges(const DnCam& c, const float* __restrict__ depth, int u, int v, Vec3& a, Vec3& b)
{
    const size_t row = (size_t)v * c.W;
    const float dr = depth[row + u + 1], dl = depth[row + u - 1], dt = depth[row - c.W + u], db = depth[row + c.W + u];
    const Vec3 pr = dn_point(c, u + 1, v, dr), pl = dn_point(c, u - 1, v, dl), pt = dn_point(c, u, v - 1, dt), pb = dn_point(c, u, v + 1, db);
    a = {pr.x - pl.x, pr.y - pl.y, pr.z - pl.z};
    b = {pt.x - pb.x, pt.y - pb.y, pt.z - pb.z};
}

__global__ void __launch_bounds__(256) depth_normal_fwd_kernel(DnCam c, const float* __restrict__ depth, float* __restrict__ out)
{
    const int u = blockIdx.x * 64 + (threadIdx.x & 63), v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= c.W || v >= c.H) return;
    const size_t HW = (size_t)c.W * c.H, pix = (size_t)v * c.W + u;
    Vec3 o = {0.f, 0.f, 0.f};
    if (u >= 1 && v >= 1 && u < c.W - 1 && v < c.H - 1) {
        Vec3 a, b;
        dn_edges(c, depth, u, v, a, b);
        const Vec3 cr = cross3(a, b);
        const float s = fmaxf(sqrtf(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z), 1e-12f);
        const Vec3 n = {cr.x / s, cr.y / s, cr.z / s};
        const float L = sqrtf(n.x * n.x + n.y * n.y + n.z * n.z) + 1e-8f;
        o = {n.x / L, n.y / L, n.z / L};
    }
    out[pix] = o.x; out[HW + pix] = o.y; out[2 * HW + pix] = o.z;
}

// dL/da and dL/db of pixel (u, v) (zero outside the interior) from the incoming gradient of its normal
__device__ __forceinline__ void dn_grad_edges(const DnCam& c, const float* __restrict__ depth, const float* __restrict__ g, int u, int v, Vec3& ga, Vec3& gb)
{
    ga = {0.f, 0.f, 0.f}; gb = {0.f, 0.f, 0.f};
    if (!(u >= 1 && v >= 1 && u < c.W - 1 && v < c.H - 1)) return;
    const size_t HW = (size_t)c.W * c.H, pix = (size_t)v * c.W + u;
    const Vec3 go = {g[pix], g[HW + pix], g[2 * HW + pix]};
    Vec3 a, b;
    dn_edges(c, depth, u, v, a, b);
    const Vec3 cr = cross3(a, b);
    const float len = sqrtf(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z);
    const float s = fmaxf(len, 1e-12f);
    const Vec3 n = {cr.x / s, cr.y / s, cr.z / s};
    const float Ln = sqrtf(n.x * n.x + n.y * n.y + n.z * n.z), Le = Ln + 1e-8f;
    // out = n / (|n| + eps):  dL/dn = g / Le - n (n . g) / (|n| Le^2)          (|n| = 0 only for a zero cross product: then the second term vanishes with n)
    const float ng = n.x * go.x + n.y * go.y + n.z * go.z;
    const float k2 = Ln > 0.f ? ng / (Ln * Le * Le) : 0.f;
    const Vec3 gn = {go.x / Le - n.x * k2, go.y / Le - n.y * k2, go.z / Le - n.z * k2};
    // n = c / max(|c|, 1e-12):  above the clamp dL/dc = (gn - n (n . gn)) / |c|, below it gn / 1e-12
    Vec3 gc;
    if (len > 1e-12f) { const float t = n.x * gn.x + n.y * gn.y + n.z * gn.z; gc = {(gn.x - n.x * t) / s, (gn.y - n.y * t) / s, (gn.z - n.z * t) / s}; }
    else gc = {gn.x / s, gn.y / s, gn.z / s};
    ga = cross3(b, gc);          // c = a x b:  dL/da = b x dL/dc,  dL/db = dL/dc x a
    gb = cross3(gc, a);
}

// One workgroup per tile of 64 x 8 pixels.  Round 6: until then every pixel recomputed dL/da, dL/db of its four neighbours (dn_grad_edges: ~300 instructions with its
// three square roots and a dozen IEEE divisions) -- four evaluations per pixel, 75 us for a 1080p image, VALU-bound at 5 x its forward.  Now a pixel's pair is
// evaluated ONCE, parked in LDS, and read by the four neighbours that need it: 512 own pixels + 16 (left / right halo, dL/da only matters) + 128 (top / bottom halo,
// dL/db) = 1.28 evaluations per pixel.  Same expression per evaluation, same order of the four terms: the gradient is what it was.
constexpr int DN_TW = 64, DN_TH = 8;
__global__ void __launch_bounds__(256) depth_normal_bwd_kernel(DnCam c, const float* __restrict__ depth, const float* __restrict__ g, float* __restrict__ gdepth)
{
    __shared__ float s_ga[3][DN_TH][DN_TW + 2];          // dL/da of pixels (u0 - 1 .. u0 + 64, v0 .. v0 + 7)
    __shared__ float s_gb[3][DN_TH + 2][DN_TW];          // dL/db of pixels (u0 .. u0 + 63, v0 - 1 .. v0 + 8)
    const int u0 = blockIdx.x * DN_TW, v0 = blockIdx.y * DN_TH, tid = threadIdx.x;
    constexpr int OWN = DN_TW * DN_TH, HALO_A = 2 * DN_TH, HALO_B = 2 * DN_TW;
    for (int j = tid; j < OWN + HALO_A + HALO_B; j += 256) {
        int lx, ly;          // tile coordinates; -1 / DN_TW and -1 / DN_TH are the halo
        if (j < OWN) { lx = j % DN_TW; ly = j / DN_TW; }
        else if (j < OWN + HALO_A) { const int k = j - OWN; ly = k >> 1; lx = (k & 1) ? DN_TW : -1; }
        else { const int k = j - OWN - HALO_A; lx = k % DN_TW; ly = (k / DN_TW) ? DN_TH : -1; }
        Vec3 ga, gb;
        dn_grad_edges(c, depth, g, u0 + lx, v0 + ly, ga, gb);          // (zero outside the image's interior)
        if (ly >= 0 && ly < DN_TH) { s_ga[0][ly][lx + 1] = ga.x; s_ga[1][ly][lx + 1] = ga.y; s_ga[2][ly][lx + 1] = ga.z; }
        if (lx >= 0 && lx < DN_TW) { s_gb[0][ly + 1][lx] = gb.x; s_gb[1][ly + 1][lx] = gb.y; s_gb[2][ly + 1][lx] = gb.z; }
    }
    __syncthreads();
    for (int j = tid; j < OWN; j += 256) {
        const int lx = j % DN_TW, ly = j / DN_TW, u = u0 + lx, v = v0 + ly;
        if (u >= c.W || v >= c.H) continue;
        float rx, ry;
        dn_ray(c, u, v, rx, ry);
        float acc = 0.f;
        // this pixel is the RIGHT point of (u - 1, v), the LEFT point of (u + 1, v), the TOP point of (u, v + 1), the BOTTOM point of (u, v - 1)
        if (u >= 1)       acc += s_ga[0][ly][lx] * rx + s_ga[1][ly][lx] * ry + s_ga[2][ly][lx];
        if (u + 1 < c.W)  acc -= s_ga[0][ly][lx + 2] * rx + s_ga[1][ly][lx + 2] * ry + s_ga[2][ly][lx + 2];
        if (v + 1 < c.H)  acc += s_gb[0][ly + 2][lx] * rx + s_gb[1][ly + 2][lx] * ry + s_gb[2][ly + 2][lx];
        if (v >= 1)       acc -= s_gb[0][ly][lx] * rx + s_gb[1][ly][lx] * ry + s_gb[2][ly][lx];
        gdepth[(size_t)v * c.W + u] = acc;
    }
}

static bool dn_cam(DnCam& c, int W, int H, float fx, float fy, float cx, float cy)
{
    if (W < 2 || H < 2 || !(fx != 0.f) || !(fy != 0.f)) return false;
    c.ifx = 1.0f / fx; c.ify = 1.0f / fy; c.mcx = -cx / fx; c.mcy = -cy / fy;
    c.wm1 = (float)(W - 1); c.hm1 = (float)(H - 1); c.W = W; c.H = H;
    return true;
}

}  // namespace ibgs

using namespace ibgs;

extern "C" {

int32_t ibgs_depth_normal_forward(void* stream, int32_t W, int32_t H, float fx, float fy, float cx, float cy, const float* depth, float* normal)
{
    DnCam c;
    if (!dn_cam(c, W, H, fx, fy, cx, cy) || !depth || !normal) { set_error("depth_normal: bad size / intrinsics / null pointer"); return -IBGS_ERR_INVALID; }
    hipLaunchKernelGGL(depth_normal_fwd_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), c, depth, normal);
    IBGS_HIP(hipGetLastError());
    return 0;
}

int32_t ibgs_depth_normal_backward(void* stream, int32_t W, int32_t H, float fx, float fy, float cx, float cy, const float* depth,
                                   const float* dL_dnormal, float* dL_ddepth)
{
    DnCam c;
    if (!dn_cam(c, W, H, fx, fy, cx, cy) || !depth || !dL_dnormal || !dL_ddepth) { set_error("depth_normal backward: bad size / intrinsics / null pointer"); return -IBGS_ERR_INVALID; }
    hipLaunchKernelGGL(depth_normal_bwd_kernel, dim3((W + DN_TW - 1) / DN_TW, (H + DN_TH - 1) / DN_TH), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), c, depth, dL_dnormal, dL_ddepth);
    IBGS_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
