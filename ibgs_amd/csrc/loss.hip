// The photometric L1 term of the reference's training step, `l1_loss(image, gt) = |image - gt|.mean()` (utils/loss_utils.py:23-24,
// train.py:302), with its gradient in the same pass: d/dx mean|x - y| = sign(x - y) / N.  torch builds the same thing out of six
// small kernels (sub, abs, mean; sign, mul, div: ~70 us for a 1080p image on MI355X, 4 % of a C3 step); here one kernel reads x and y
// once, writes sign(x - y) / N and per-workgroup partial sums, a second (one workgroup) adds the partials in a fixed order -- no float
// atomics, the value is reproducible.  HBM-bound: 12 bytes per element.
#include "common.h"

namespace ibgs {

constexpr int L1_THREADS = 256, L1_MAX_BLOCKS = 1024;

__global__ void __launch_bounds__(L1_THREADS) l1_partial_kernel(const float* __restrict__ x, const float* __restrict__ y, size_t n, float inv_n,
                                                                float* __restrict__ grad /* may be null */, double* __restrict__ partial)
{
    __shared__ double s_w[L1_THREADS / 64];
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(grad)) & 15u) == 0;
    float acc = 0.f;
    auto one = [&](float a, float b) -> float {
        const float d = a - b;
        acc += fabsf(d);
        return d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);          // sign(0) = 0, as torch.sign
    };
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t i = (size_t)blockIdx.x * L1_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * L1_THREADS) {
        const float4 a = reinterpret_cast<const float4*>(x)[i], b = reinterpret_cast<const float4*>(y)[i];
        const float4 g = make_float4(one(a.x, b.x), one(a.y, b.y), one(a.z, b.z), one(a.w, b.w));
        if (grad) reinterpret_cast<float4*>(grad)[i] = g;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * L1_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * L1_THREADS) {
        const float g = one(x[i], y[i]);
        if (grad) grad[i] = g;
    }
    double v = (double)acc;
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < L1_THREADS / 64; w++) t += s_w[w]; partial[blockIdx.x] = t; }
}

// the gradient on its own, scaled by a DEVICE scalar (autograd's incoming gradient): grad = sign(x - y) * (*scale) / n, one pass
__global__ void __launch_bounds__(L1_THREADS) l1_grad_kernel(const float* __restrict__ x, const float* __restrict__ y, size_t n, float inv_n,
                                                             const float* __restrict__ scale, float* __restrict__ grad)
{
    const float k = inv_n * (scale ? *scale : 1.0f);
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(grad)) & 15u) == 0;
    auto one = [&](float a, float b) -> float { const float d = a - b; return d > 0.f ? k : (d < 0.f ? -k : 0.f * k); };
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t i = (size_t)blockIdx.x * L1_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * L1_THREADS) {
        const float4 a = reinterpret_cast<const float4*>(x)[i], b = reinterpret_cast<const float4*>(y)[i];
        reinterpret_cast<float4*>(grad)[i] = make_float4(one(a.x, b.x), one(a.y, b.y), one(a.z, b.z), one(a.w, b.w));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * L1_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * L1_THREADS) grad[i] = one(x[i], y[i]);
}

// The gradient that the loss pass stored (sign(x - y) / n) times autograd's incoming gradient, a DEVICE scalar, in place.  A loss term that enters the
// total with weight one (`loss = l1 + ...`) arrives with 1.0: every workgroup reads the scalar and leaves -- the backward of the term then moves no data.
__global__ void __launch_bounds__(L1_THREADS) l1_rescale_kernel(float* __restrict__ grad, size_t n, const float* __restrict__ scale)
{
    const float k = *scale;
    if (k == 1.0f) return;          // (wave-uniform: a scalar load and a branch)
    const bool vec = (reinterpret_cast<uintptr_t>(grad) & 15u) == 0;
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t i = (size_t)blockIdx.x * L1_THREADS + threadIdx.x; i < n4; i += (size_t)gridDim.x * L1_THREADS) {
        float4 g = reinterpret_cast<float4*>(grad)[i];
        g.x *= k; g.y *= k; g.z *= k; g.w *= k;
        reinterpret_cast<float4*>(grad)[i] = g;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * L1_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * L1_THREADS) grad[i] *= k;
}

__global__ void __launch_bounds__(64) l1_final_kernel(const double* __restrict__ partial, int nblocks, double inv_n, float* __restrict__ loss)
{
    double v = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 64) v += partial[i];
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (threadIdx.x == 0) *loss = (float)(v * inv_n);
}

}  // namespace ibgs

using namespace ibgs;

extern "C" size_t ibgs_required_l1(void) { return sizeof(double) * L1_MAX_BLOCKS + 128; }

extern "C" int32_t ibgs_l1_loss(void* stream, int64_t n, const float* x, const float* y, float* grad, float* loss, char* scratch, size_t scratch_bytes)
{
    if (n <= 0 || !x || !y || !loss) { set_error("ibgs_l1_loss: n > 0 and x, y, loss required"); return -IBGS_ERR_INVALID; }
    if (!scratch || scratch_bytes < ibgs_required_l1()) { set_error("ibgs_l1_loss: scratch too small"); return -IBGS_ERR_ALLOC; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    double* partial = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(scratch) + 127) & ~uintptr_t(127));
    const size_t per_block = (size_t)L1_THREADS * 4 * 4;          // ~4 quads per thread
    size_t nb = ((size_t)n + per_block - 1) / per_block;
    if (nb > (size_t)L1_MAX_BLOCKS) nb = L1_MAX_BLOCKS;
    const double inv_n = 1.0 / (double)n;
    hipLaunchKernelGGL(l1_partial_kernel, dim3((unsigned)nb), dim3(L1_THREADS), 0, s, x, y, (size_t)n, (float)inv_n, grad, partial);
    IBGS_HIP(hipGetLastError());
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(64), 0, s, partial, (int)nb, inv_n, loss);
    IBGS_HIP(hipGetLastError());
    return 0;
}

extern "C" int32_t ibgs_l1_grad(void* stream, int64_t n, const float* x, const float* y, const float* scale_dev, float* grad)
{
    if (n <= 0 || !x || !y || !grad) { set_error("ibgs_l1_grad: n > 0 and x, y, grad required"); return -IBGS_ERR_INVALID; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t per_block = (size_t)L1_THREADS * 4 * 4;
    size_t nb = ((size_t)n + per_block - 1) / per_block;
    if (nb > (size_t)L1_MAX_BLOCKS * 4) nb = (size_t)L1_MAX_BLOCKS * 4;
    hipLaunchKernelGGL(l1_grad_kernel, dim3((unsigned)nb), dim3(L1_THREADS), 0, s, x, y, (size_t)n, (float)(1.0 / (double)n), scale_dev, grad);
    IBGS_HIP(hipGetLastError());
    return 0;
}

extern "C" int32_t ibgs_l1_rescale(void* stream, int64_t n, float* grad, const float* scale_dev)
{
    if (n <= 0 || !grad || !scale_dev) { set_error("ibgs_l1_rescale: n > 0, grad and scale_dev required"); return -IBGS_ERR_INVALID; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t per_block = (size_t)L1_THREADS * 4 * 4;
    size_t nb = ((size_t)n + per_block - 1) / per_block;
    if (nb > (size_t)L1_MAX_BLOCKS * 4) nb = (size_t)L1_MAX_BLOCKS * 4;
    hipLaunchKernelGGL(l1_rescale_kernel, dim3((unsigned)nb), dim3(L1_THREADS), 0, s, grad, (size_t)n, scale_dev);
    IBGS_HIP(hipGetLastError());
    return 0;
}
