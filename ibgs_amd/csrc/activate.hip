// Row G of SURVEY.md 8(a), the model side of the render glue: the three activations every render() call applies to the raw parameters
// (scene/gaussian_model.py:44-52, 128-147: `get_scaling` = exp(_scaling), `get_rotation` = F.normalize(_rotation), `get_opacity` = sigmoid(_opacity)),
// forward and backward, one kernel each way.
//
// In torch that is exp, sigmoid, and for the normalisation a norm reduction, a clamp, an expand and a division -- and their mirror images in the
// backward: ~17 launches of 4-19 us each over arrays of 4-16 bytes per Gaussian (profiles/r05_train_iter_kernels.txt: ~0.09 ms of a 2.0 ms trainer
// iteration at 1 M Gaussians).  Here: 32 B read + 32 B written per Gaussian forward, 64 B + 32 B backward; a wave handles 64 Gaussians, every array is
// read and written with consecutive lanes on consecutive words (the (P, 3) and (P, 4) arrays as flat word streams, the per-Gaussian values exchanged
// through LDS), so every access is a full cache line.
//
// Arithmetic = torch's, operation by operation: exp -> expf; sigmoid -> 1 / (1 + expf(-x)); normalize -> x / max(sqrt(sum x^2), 1e-12) (F.normalize, p = 2,
// eps = 1e-12).  Backward: dL/dx = g y (exp); g (1 - y) y (sigmoid_backward); for the normalisation, with n = |x|, d = max(n, eps):
// dL/dx = g / d - x (g . x) / (d^2 n) where n >= eps (the clamp passes the gradient), g / d below it -- autograd's own chain through div, expand, clamp_min, norm.
#include "common.h"

namespace ibgs {

constexpr float NORMALIZE_EPS = 1e-12f;

__global__ void __launch_bounds__(256) activate_fwd_kernel(int P, const float* __restrict__ raw_scale, const float* __restrict__ raw_rot, const float* __restrict__ raw_opacity,
                                                           float* __restrict__ scale, float* __restrict__ rot, float* __restrict__ opacity)
{
#pragma clang fp contract(off)
    __shared__ float s_q[4][64 * 4 + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g0 = (blockIdx.x * 4 + wave) * 64;          // first Gaussian of this wave
    if (g0 >= P) return;
    const int n = min(64, P - g0);
    // scales: 3 n consecutive words, elementwise
    if (raw_scale) {
        for (int k = lane; k < 3 * n; k += 64) scale[(size_t)3 * g0 + k] = expf(raw_scale[(size_t)3 * g0 + k]);
    }
    if (raw_opacity && lane < n) opacity[g0 + lane] = 1.0f / (1.0f + expf(-raw_opacity[g0 + lane]));
    if (raw_rot) {
        float* q = s_q[wave];
        for (int k = lane; k < 4 * n; k += 64) q[k] = raw_rot[(size_t)4 * g0 + k];          // (one wave: LDS operations execute in order)
        if (lane < n) {
            const float4 v = *reinterpret_cast<const float4*>(q + 4 * lane);
            const float d = fmaxf(sqrtf(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w), NORMALIZE_EPS);
            *reinterpret_cast<float4*>(q + 4 * lane) = make_float4(v.x / d, v.y / d, v.z / d, v.w / d);
        }
        for (int k = lane; k < 4 * n; k += 64) rot[(size_t)4 * g0 + k] = q[k];
    }
}

__global__ void __launch_bounds__(256) activate_bwd_kernel(int P, const float* __restrict__ raw_scale, const float* __restrict__ raw_rot, const float* __restrict__ raw_opacity,
                                                           const float* __restrict__ g_scale, const float* __restrict__ g_rot, const float* __restrict__ g_opacity,
                                                           float* __restrict__ d_scale, float* __restrict__ d_rot, float* __restrict__ d_opacity)
{
#pragma clang fp contract(off)
    __shared__ float s_q[4][2][64 * 4 + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g0 = (blockIdx.x * 4 + wave) * 64;
    if (g0 >= P) return;
    const int n = min(64, P - g0);
    if (d_scale) {
        for (int k = lane; k < 3 * n; k += 64) d_scale[(size_t)3 * g0 + k] = g_scale[(size_t)3 * g0 + k] * expf(raw_scale[(size_t)3 * g0 + k]);
    }
    if (d_opacity && lane < n) {
        const float y = 1.0f / (1.0f + expf(-raw_opacity[g0 + lane]));
        d_opacity[g0 + lane] = g_opacity[g0 + lane] * (1.0f - y) * y;
    }
    if (d_rot) {
        float* q = s_q[wave][0]; float* gq = s_q[wave][1];
        for (int k = lane; k < 4 * n; k += 64) { q[k] = raw_rot[(size_t)4 * g0 + k]; gq[k] = g_rot[(size_t)4 * g0 + k]; }
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < n) {
            const float4 v = *reinterpret_cast<const float4*>(q + 4 * lane), g = *reinterpret_cast<const float4*>(gq + 4 * lane);
            const float nn = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
            const float d = fmaxf(nn, NORMALIZE_EPS);
            out = make_float4(g.x / d, g.y / d, g.z / d, g.w / d);
            if (nn >= NORMALIZE_EPS && nn > 0.f) {
                // through the denominator: dL/dd = -(g . x) / d^2, d = n here, dn/dx = x / n
                const float c = (g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w) / (d * d) / nn;
                out.x -= v.x * c; out.y -= v.y * c; out.z -= v.z * c; out.w -= v.w * c;
            }
        }
        if (lane < n) *reinterpret_cast<float4*>(q + 4 * lane) = out;
        for (int k = lane; k < 4 * n; k += 64) d_rot[(size_t)4 * g0 + k] = q[k];
    }
}

}  // namespace ibgs

using namespace ibgs;

extern "C" {

int32_t ibgs_activate_forward(void* stream, int32_t P, const float* raw_scale, const float* raw_rot, const float* raw_opacity, float* scale, float* rot, float* opacity)
{
    if (P < 0 || (raw_scale && !scale) || (raw_rot && !rot) || (raw_opacity && !opacity)) { set_error("activate: bad size / output missing for a given input"); return -IBGS_ERR_INVALID; }
    if (P == 0) return 0;
    hipLaunchKernelGGL(activate_fwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P, raw_scale, raw_rot, raw_opacity, scale, rot, opacity);
    IBGS_HIP(hipGetLastError());
    return 0;
}

int32_t ibgs_activate_backward(void* stream, int32_t P, const float* raw_scale, const float* raw_rot, const float* raw_opacity,
                               const float* g_scale, const float* g_rot, const float* g_opacity, float* d_scale, float* d_rot, float* d_opacity)
{
    if (P < 0 || (d_scale && (!raw_scale || !g_scale)) || (d_rot && (!raw_rot || !g_rot)) || (d_opacity && (!raw_opacity || !g_opacity))) {
        set_error("activate backward: an output without its raw input / incoming gradient"); return -IBGS_ERR_INVALID;
    }
    if (P == 0) return 0;
    hipLaunchKernelGGL(activate_bwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P, raw_scale, raw_rot, raw_opacity,
                       g_scale, g_rot, g_opacity, d_scale, d_rot, d_opacity);
    IBGS_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
