// The Adam update as both optimiser kernels apply it (adam.hip: one launch over dense gradients; preprocess_bwd.hip: the SH coefficients straight from the
// per-view factors, without a dense dL/dsh in memory).  One definition, so that the two paths cannot drift: tests/test_gpu_adam.py asserts bit-identical
// parameters and moments.  Same update rule and operation order as torch.optim.Adam (no weight decay, no amsgrad):
//   m = m + (1 - b1) (g - m);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#pragma once
#include <cmath>
#include "../../include/ibgs_rast.h"

namespace ibgs {

// per-tensor constants as the kernels use them: every scalar is rounded to fp32 exactly once, like torch's kernels do
struct AdamDev { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; long long numel; float b2, omb1, omb2, step_size, inv_bc2_sqrt, eps; };

inline AdamDev adam_dev_from(const ibgs_adam_tensor& d)
{
    AdamDev o;
    o.param = d.param; o.grad = d.grad; o.exp_avg = d.exp_avg; o.exp_avg_sq = d.exp_avg_sq; o.numel = d.numel;
    o.b2 = (float)d.beta2; o.omb1 = (float)(1.0 - d.beta1); o.omb2 = (float)(1.0 - d.beta2);
    o.step_size = (float)(d.lr / d.bias_correction1); o.inv_bc2_sqrt = (float)(1.0 / sqrt(d.bias_correction2)); o.eps = (float)d.eps;
    return o;
}

// (no contraction: which products the compiler fuses into an fma depends on the code around the call, and the two kernels that share this function must round alike)
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamDev& d)
{
#pragma clang fp contract(off)
    const float step_size = d.step_size, eps = d.eps;
    m = m + d.omb1 * (g - m);                      // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = d.b2 * v + d.omb2 * g * g;                 // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) * d.inv_bc2_sqrt + eps;
    p = p - step_size * (m / denom);               // param.addcdiv_(exp_avg, denom, value = -step_size)
}

}  // namespace ibgs
