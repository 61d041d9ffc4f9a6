// the optimiser step of the reference trainer (train.py:421-430; torch.optim.Adam over the eight
// Gaussian parameter groups, scene/gaussian_model.py:227-241) as ONE launch over all groups.  Same update rule and
// operation order as torch.optim.Adam (no weight decay, no amsgrad):
//   m = m + (1 - b1) (g - m);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// HBM-bound: 7 floats of traffic per parameter (p, g, m, v read; p, m, v written); 16-byte vector accesses.
#include <cmath>
#include "common.h"
#include "adam_math.h"          // AdamDev, adam_one: shared with the SH-from-factors step (preprocess_bwd.hip)

namespace ibgs {

struct AdamTable { AdamDev t[IBGS_ADAM_MAX_TENSORS]; unsigned long long first_block[IBGS_ADAM_MAX_TENSORS + 1]; int n; };

constexpr int ADAM_THREADS = 256, ADAM_VEC = 4, ADAM_ITEMS = 4;            // 4096 floats per workgroup
constexpr size_t ADAM_CHUNK = (size_t)ADAM_THREADS * ADAM_VEC * ADAM_ITEMS;

__global__ void __launch_bounds__(ADAM_THREADS) adam_kernel(AdamTable tab)
{
    // which tensor does this workgroup belong to (<= 16 entries, wave-uniform)
    int k = 0;
    while (k + 1 < tab.n && (unsigned long long)blockIdx.x >= tab.first_block[k + 1]) k++;
    const AdamDev d = tab.t[k];
    const size_t base = ((size_t)blockIdx.x - (size_t)tab.first_block[k]) * ADAM_CHUNK;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(d.param) | reinterpret_cast<uintptr_t>(d.grad) | reinterpret_cast<uintptr_t>(d.exp_avg) |
                          reinterpret_cast<uintptr_t>(d.exp_avg_sq)) & 15u) == 0;
#pragma unroll
    for (int it = 0; it < ADAM_ITEMS; it++) {
        const size_t i = base + ((size_t)it * ADAM_THREADS + threadIdx.x) * ADAM_VEC;
        if (i >= (size_t)d.numel) break;
        if (vec_ok && i + ADAM_VEC <= (size_t)d.numel) {
            float4 p = *reinterpret_cast<float4*>(d.param + i);
            const float4 g = *reinterpret_cast<const float4*>(d.grad + i);
            float4 m = *reinterpret_cast<float4*>(d.exp_avg + i), v = *reinterpret_cast<float4*>(d.exp_avg_sq + i);
            adam_one(p.x, g.x, m.x, v.x, d);
            adam_one(p.y, g.y, m.y, v.y, d);
            adam_one(p.z, g.z, m.z, v.z, d);
            adam_one(p.w, g.w, m.w, v.w, d);
            *reinterpret_cast<float4*>(d.param + i) = p;
            *reinterpret_cast<float4*>(d.exp_avg + i) = m; *reinterpret_cast<float4*>(d.exp_avg_sq + i) = v;
        } else {
            for (size_t j = i; j < i + ADAM_VEC && j < (size_t)d.numel; j++) {
                float p = d.param[j], m = d.exp_avg[j], v = d.exp_avg_sq[j];
                adam_one(p, d.grad[j], m, v, d);
                d.param[j] = p; d.exp_avg[j] = m; d.exp_avg_sq[j] = v;
            }
        }
    }
}

}  // namespace ibgs

extern "C" int32_t ibgs_adam_step(void* stream, int32_t n_tensors, const ibgs_adam_tensor* tensors)
{
    using namespace ibgs;
    if (n_tensors <= 0) return 0;
    if (n_tensors > IBGS_ADAM_MAX_TENSORS || !tensors) { set_error("ibgs_adam_step: 1..%d tensors per call", IBGS_ADAM_MAX_TENSORS); return -IBGS_ERR_INVALID; }
    AdamTable tab;
    tab.n = 0;
    unsigned long long blocks = 0;
    for (int k = 0; k < n_tensors; k++) {
        const ibgs_adam_tensor& d = tensors[k];
        if (d.numel <= 0) continue;
        if (!d.param || !d.grad || !d.exp_avg || !d.exp_avg_sq) { set_error("ibgs_adam_step: null pointer in tensor %d", k); return -IBGS_ERR_INVALID; }
        if (!(d.bias_correction1 > 0.f) || !(d.bias_correction2 > 0.f)) { set_error("ibgs_adam_step: bias corrections must be positive"); return -IBGS_ERR_INVALID; }
        tab.t[tab.n] = adam_dev_from(d);
        tab.first_block[tab.n] = blocks; tab.n++;
        blocks += ((unsigned long long)d.numel + ADAM_CHUNK - 1) / ADAM_CHUNK;
    }
    if (tab.n == 0) return 0;
    tab.first_block[tab.n] = blocks;
    if (blocks > 0x7FFFFFFFull) { set_error("ibgs_adam_step: too many elements"); return -IBGS_ERR_INVALID; }
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(ADAM_THREADS), 0, reinterpret_cast<hipStream_t>(stream), tab);
    IBGS_HIP(hipGetLastError());
    return 0;
}
