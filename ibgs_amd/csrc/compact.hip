// densification surgery of the reference trainer as ONE data-movement pass.
//
// The reference prunes / extends its point set with boolean indexing + torch.cat per tensor, per optimiser state:
//   _prune_optimizer (scene/gaussian_model.py:377-395): param[mask], exp_avg[mask], exp_avg_sq[mask] for each of 8 groups,
//   cat_tensors_to_optimizer (:423-444): cat(param, new), cat(exp_avg, zeros), cat(exp_avg_sq, zeros),
//   prune_points (:397-421) additionally masks six per-point statistics tensors,
// i.e. ~30 gather kernels + ~24 concatenations (each with its own allocation) every 100 iterations, three times per
// densification event (densify_and_clone, densify_and_split, prune; :580-597).  Here: one exclusive scan of the keep mask
// (ibgs_compact_plan, the only host sync: the caller needs the new row count to allocate), then ONE launch moves every
// tensor (ibgs_compact_apply): kept rows in their original order, then the appended rows (copied, or zeros for the Adam
// moments of new points).  Pure data movement: results are bit-identical to the torch formulation.
// HBM-bound: 8 bytes per surviving float.
#include "common.h"

namespace ibgs {

struct CompactTable { ibgs_compact_tensor t[IBGS_COMPACT_MAX_TENSORS]; int n; };

constexpr int COMPACT_THREADS = 256;

__global__ void __launch_bounds__(COMPACT_THREADS) mask_to_u32_kernel(const uint8_t* __restrict__ mask, uint32_t* __restrict__ out, int n)
{
    const int i = blockIdx.x * COMPACT_THREADS + threadIdx.x;
    if (i < n) out[i] = mask ? (mask[i] != 0 ? 1u : 0u) : 1u;
}

// grid.y = tensor, grid.x strides over the elements of (old rows) then (appended rows)
__global__ void __launch_bounds__(COMPACT_THREADS) compact_apply_kernel(CompactTable tab, const uint32_t* __restrict__ offsets, int n_old, int n_app)
{
    const ibgs_compact_tensor d = tab.t[blockIdx.y];
    const uint32_t w = (uint32_t)d.width;
    const uint32_t n_keep = offsets[n_old];
    const size_t old_elems = (size_t)n_old * w, app_elems = (size_t)n_app * w;
    const size_t stride = (size_t)gridDim.x * COMPACT_THREADS;
    for (size_t e = (size_t)blockIdx.x * COMPACT_THREADS + threadIdx.x; e < old_elems; e += stride) {
        const uint32_t row = (uint32_t)(e / w), col = (uint32_t)(e - (size_t)row * w);
        const uint32_t o = offsets[row];
        if (offsets[row + 1] != o) d.dst[(size_t)o * w + col] = d.src[e];           // kept rows: offsets step by one
    }
    float* tail = d.dst + (size_t)n_keep * w;
    for (size_t e = (size_t)blockIdx.x * COMPACT_THREADS + threadIdx.x; e < app_elems; e += stride)
        tail[e] = d.append ? d.append[e] : 0.0f;
}


// The densification statistics of one training view (train.py:400-405, scene/gaussian_model.py:600-604) in ONE kernel: for the Gaussians the view saw
// (radii > 0): max_radii2D = max(max_radii2D, radius), accum += |dL/dmean2D.xy|, accum_abs += |dL/dmean2D_abs.xy|, both denominators += 1.  The reference
// writes them as five boolean-indexed updates (`t[mask] = ...`, `t[mask] += ...`): each one a nonzero + gather + scatter and a HOST SYNC for the index count.
__global__ void __launch_bounds__(256) densify_stats_kernel(int P, const int32_t* __restrict__ radii, const float* __restrict__ g2, const float* __restrict__ g2a,
                                                            float* __restrict__ accum, float* __restrict__ accum_abs, float* __restrict__ denom,
                                                            float* __restrict__ denom_abs, float* __restrict__ max_radii)
{
#pragma clang fp contract(off)          // the norms as torch.norm forms them: sqrt(x * x + y * y), no fused multiply-add
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    if (r <= 0) return;
    if (max_radii) max_radii[i] = fmaxf(max_radii[i], (float)r);
    if (accum && g2) { const float x = g2[3 * i], y = g2[3 * i + 1]; accum[i] += sqrtf(x * x + y * y); }
    if (accum_abs && g2a) { const float x = g2a[3 * i], y = g2a[3 * i + 1]; accum_abs[i] += sqrtf(x * x + y * y); }
    if (denom) denom[i] += 1.0f;
    if (denom_abs) denom_abs[i] += 1.0f;
}

}  // namespace ibgs

using namespace ibgs;

extern "C" {

size_t ibgs_required_compact(int32_t n_old)
{
    const size_t n = (size_t)(n_old > 0 ? n_old : 0);
    return (n + 1 + scan_scratch_elems(n + 1) + 64) * sizeof(uint32_t) + 256;
}

int64_t ibgs_compact_plan(void* stream, int32_t n_old, const uint8_t* keep_mask, char* scratch, size_t scratch_bytes)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (n_old < 0) { set_error("n_old < 0"); return -IBGS_ERR_INVALID; }
    if (n_old == 0) return 0;
    if (!scratch || scratch_bytes < ibgs_required_compact(n_old)) { set_error("compact scratch too small"); return -IBGS_ERR_ALLOC; }
    uint32_t* off = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(scratch) + 127) & ~uintptr_t(127));
    uint32_t* scan_scratch = off + n_old + 1 + 31;
    hipLaunchKernelGGL(mask_to_u32_kernel, dim3((n_old + COMPACT_THREADS - 1) / COMPACT_THREADS), dim3(COMPACT_THREADS), 0, s, keep_mask, off, n_old);
    IBGS_HIP(hipGetLastError());
    int rc = exclusive_scan_u32(s, off, off, (size_t)n_old, scan_scratch, scan_scratch_elems((size_t)n_old + 1), true);
    if (rc) return rc;
    uint32_t total = 0;
    IBGS_HIP(hipMemcpyAsync(&total, off + n_old, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    IBGS_HIP(hipStreamSynchronize(s));
    return (int64_t)total;
}

int32_t ibgs_compact_apply(void* stream, int32_t n_tensors, const ibgs_compact_tensor* tensors, int32_t n_old, int32_t n_app,
                           const char* scratch)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (n_tensors <= 0) return 0;
    if (n_tensors > IBGS_COMPACT_MAX_TENSORS) { set_error("at most %d tensors per call", IBGS_COMPACT_MAX_TENSORS); return -IBGS_ERR_INVALID; }
    if (n_old < 0 || n_app < 0 || !tensors || (n_old > 0 && !scratch)) { set_error("bad compact arguments"); return -IBGS_ERR_INVALID; }
    if (n_old == 0 && n_app == 0) return 0;
    CompactTable tab; tab.n = n_tensors;
    size_t most = 0;
    for (int i = 0; i < n_tensors; i++) {
        const ibgs_compact_tensor& t = tensors[i];
        if (t.width <= 0 || !t.dst || (n_old > 0 && !t.src)) { set_error("compact tensor %d: bad descriptor", i); return -IBGS_ERR_INVALID; }
        tab.t[i] = t;
        const size_t el = (size_t)(n_old > n_app ? n_old : n_app) * (size_t)t.width;
        most = el > most ? el : most;
    }
    if (n_old == 0) { set_error("n_old == 0: nothing to compact (dst is the appended rows)"); return -IBGS_ERR_INVALID; }
    const uint32_t* off = reinterpret_cast<const uint32_t*>((reinterpret_cast<uintptr_t>(scratch) + 127) & ~uintptr_t(127));
    size_t blocks = (most + COMPACT_THREADS * 4 - 1) / (COMPACT_THREADS * 4);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(compact_apply_kernel, dim3((unsigned)blocks, (unsigned)n_tensors), dim3(COMPACT_THREADS), 0, s, tab, off, n_old, n_app);
    IBGS_HIP(hipGetLastError());
    return 0;
}

int32_t ibgs_densify_stats(void* stream, int32_t P, const int32_t* radii, const float* dL_dmean2D, const float* dL_dmean2D_abs,
                           float* accum, float* accum_abs, float* denom, float* denom_abs, float* max_radii2D)
{
    if (P <= 0) return 0;
    if (!radii) { set_error("densify_stats: radii required"); return -IBGS_ERR_INVALID; }
    hipLaunchKernelGGL(densify_stats_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), P, radii, dL_dmean2D, dL_dmean2D_abs,
                       accum, accum_abs, denom, denom_abs, max_radii2D);
    IBGS_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
