// Section 8(f) row 3: mean squared distance to the 3 nearest neighbours (scale initialisation).
//
// Behaviour: submodules/simple-knn/simple_knn.cu:45-220 + spatial.cu:15-26 (distCUDA2): for every point the mean of
// its three smallest squared distances to the OTHER points (exact search, duplicates count with distance 0).  The
// reference prunes 1024-point Morton boxes with one thread per point; here a WAVE owns 64 Morton-consecutive points,
// boxes hold 256 points, a box is skipped by a wave-uniform ballot when no lane can improve its third-best distance,
// and a surviving box is staged once into LDS (coalesced, from a Morton-sorted copy of the points) and consumed by all
// 64 lanes as broadcast reads.  No host read-back (the reference copies min / max to the host, simple_knn.cu:195-198).
#include "common.h"
#include <cfloat>

namespace ibgs {

constexpr int KNN_BOX = 256;

struct KnnState {
    float* bounds;          // 6: min xyz, max xyz (including the origin, like the reference's reduce with init 0)
    uint32_t* codes[2];     // Morton codes (ping-pong)
    uint32_t* idx[2];       // point ids (ping-pong); idx[0] = Morton order after the sort
    float* sorted;          // P x 4 floats: xyz of the points in Morton order (+ pad)
    float* boxes;           // nboxes x 8: min xyz, pad, max xyz, pad
    uint32_t* hist; size_t hist_elems;
    static KnnState carve(char* base, size_t P, size_t* total)
    {
        Carver c(base);
        KnnState k;
        k.bounds = c.take<float>(8);
        k.codes[0] = c.take<uint32_t>(P); k.codes[1] = c.take<uint32_t>(P);
        k.idx[0] = c.take<uint32_t>(P); k.idx[1] = c.take<uint32_t>(P);
        k.sorted = c.take<float>(P * 4);
        k.boxes = c.take<float>(((P + KNN_BOX - 1) / KNN_BOX) * 8 + 8);
        k.hist_elems = radix_hist_elems(P);
        k.hist = c.take<uint32_t>(k.hist_elems);
        if (total) *total = (size_t)(c.cur - reinterpret_cast<uintptr_t>(base)) + 128;
        return k;
    }
};

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, WAVE));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, WAVE));
    return v;
}

// order-preserving float <-> uint map so that min / max can use integer atomics
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

__global__ void __launch_bounds__(256) knn_bounds_kernel(uint32_t P, const float* __restrict__ pts, uint32_t* __restrict__ ob /* 6, pre-set to the origin */)
{
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; a++) { const float v = pts[3 * i + a]; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = wave_min(mn[a]), hi = wave_max(mx[a]);
        if ((threadIdx.x & 63) == 0) { atomicMin(&ob[a], f2ord(lo)); atomicMax(&ob[3 + a], f2ord(hi)); }
    }
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ void __launch_bounds__(256) knn_morton_kernel(uint32_t P, const float* __restrict__ pts, const uint32_t* __restrict__ ob,
                                                         uint32_t* __restrict__ codes, uint32_t* __restrict__ idx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint32_t c = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = ord2f(ob[a]), hi = ord2f(ob[3 + a]);
        const float t = (pts[3 * i + a] - lo) / (hi - lo);          // simple_knn.cu:65-67 (NaN for a degenerate axis -> cell 0)
        const uint32_t q = (t > 0.f) ? min(1023u, (uint32_t)(t * 1023.0f)) : 0u;
        c |= prep_morton(q) << a;
    }
    codes[i] = c; idx[i] = i;
}

__global__ void __launch_bounds__(KNN_BOX) knn_boxes_kernel(uint32_t P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                            float4* __restrict__ sorted, float4* __restrict__ boxes)
{
    __shared__ float s_mn[4][3], s_mx[4][3];
    const uint32_t i = blockIdx.x * KNN_BOX + threadIdx.x;
    float p[3] = {0, 0, 0};
    const bool valid = i < P;
    if (valid) { const uint32_t id = order[i]; p[0] = pts[3 * id]; p[1] = pts[3 * id + 1]; p[2] = pts[3 * id + 2]; sorted[i] = make_float4(p[0], p[1], p[2], 0.f); }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = wave_min(valid ? p[a] : FLT_MAX), hi = wave_max(valid ? p[a] : -FLT_MAX);
        if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6][a] = lo; s_mx[threadIdx.x >> 6][a] = hi; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float mn[3], mx[3];
        for (int a = 0; a < 3; a++) {
            mn[a] = fminf(fminf(s_mn[0][a], s_mn[1][a]), fminf(s_mn[2][a], s_mn[3][a]));
            mx[a] = fmaxf(fmaxf(s_mx[0][a], s_mx[1][a]), fmaxf(s_mx[2][a], s_mx[3][a]));
        }
        boxes[2 * blockIdx.x] = make_float4(mn[0], mn[1], mn[2], 0.f);
        boxes[2 * blockIdx.x + 1] = make_float4(mx[0], mx[1], mx[2], 0.f);
    }
}

__device__ __forceinline__ void update3(float d, float& b0, float& b1, float& b2)
{   // updateKBest<3>, simple_knn.cu:132-145
    if (b0 > d) { const float t = b0; b0 = d; d = t; }
    if (b1 > d) { const float t = b1; b1 = d; d = t; }
    if (b2 > d) { b2 = d; }
}

__device__ __forceinline__ float box_dist2(const float4& mn, const float4& mx, float x, float y, float z)
{   // distBoxPoint, simple_knn.cu:119-129
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (x < mn.x || x > mx.x) dx = fminf(fabsf(x - mn.x), fabsf(x - mx.x));
    if (y < mn.y || y > mx.y) dy = fminf(fabsf(y - mn.y), fabsf(y - mx.y));
    if (z < mn.z || z > mx.z) dz = fminf(fabsf(z - mn.z), fabsf(z - mx.z));
    return dx * dx + dy * dy + dz * dz;
}

// One wave per 64 Morton-consecutive points.
__global__ void __launch_bounds__(64) knn_search_kernel(uint32_t P, const float4* __restrict__ sorted, const float4* __restrict__ boxes,
                                                        const uint32_t* __restrict__ order, float* __restrict__ out)
{
    __shared__ float4 s_pts[KNN_BOX];
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.x * 64 + lane;
    const bool valid = i < P;
    const float4 me = valid ? sorted[i] : make_float4(0, 0, 0, 0);
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    const uint32_t nboxes = (P + KNN_BOX - 1) / KNN_BOX;
    const uint32_t own = (blockIdx.x * 64) / KNN_BOX;
    // visit boxes outwards from the wave's own box so that the pruning bound tightens early: own, own-1, own+1, own-2, ...
    for (uint32_t step = 0; step < 2 * nboxes; step++) {
        const int off = (step & 1) ? -(int)((step + 1) >> 1) : (int)(step >> 1);
        const int b = (int)own + off;
        if (b < 0 || b >= (int)nboxes) continue;
        const float4 mn = boxes[2 * b], mx = boxes[2 * b + 1];
        const bool need = valid && !(box_dist2(mn, mx, me.x, me.y, me.z) > b2);
        if (__builtin_amdgcn_ballot_w64(need) == 0ull) continue;       // wave-uniform prune
        const uint32_t first = (uint32_t)b * KNN_BOX;
        const uint32_t cnt = min((uint32_t)KNN_BOX, P - first);
        __syncthreads();
        for (uint32_t k = lane; k < cnt; k += 64) s_pts[k] = sorted[first + k];
        __syncthreads();
        if (need) {
            for (uint32_t k = 0; k < cnt; k++) {
                if (first + k == i) continue;                            // not itself (duplicates elsewhere do count)
                const float4 q = s_pts[k];
                const float dx = q.x - me.x, dy = q.y - me.y, dz = q.z - me.z;
                update3(dx * dx + dy * dy + dz * dz, b0, b1, b2);
            }
        }
    }
    if (valid) out[order[i]] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace ibgs

using namespace ibgs;

extern "C" {

size_t ibgs_required_knn(int32_t P) { size_t t; KnnState::carve(nullptr, (size_t)(P > 0 ? P : 0), &t); return t; }

int32_t ibgs_knn_mean_dist2(void* stream, int32_t P, const float* points, float* out, char* scratch, size_t scratch_bytes)
{
    if (P <= 0) return 0;
    if (!points || !out || !scratch || scratch_bytes < ibgs_required_knn(P)) { set_error("knn: bad arguments / scratch too small"); return -IBGS_ERR_INVALID; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    KnnState k = KnnState::carve(scratch, (size_t)P, nullptr);
    // bounds start at the origin (ordered-uint encoding of 0.0f = 0x80000000), like the reference's reduce with init {0,0,0}
    uint32_t* ob = reinterpret_cast<uint32_t*>(k.bounds);
    const uint32_t zero_ord[6] = {0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u};
    IBGS_HIP(hipMemcpyAsync(ob, zero_ord, sizeof(zero_ord), hipMemcpyHostToDevice, s));
    const unsigned nb = (unsigned)((P + 255) / 256);
    hipLaunchKernelGGL(knn_bounds_kernel, dim3(nb < 1024u ? nb : 1024u), dim3(256), 0, s, (uint32_t)P, points, ob);
    hipLaunchKernelGGL(knn_morton_kernel, dim3(nb), dim3(256), 0, s, (uint32_t)P, points, ob, k.codes[0], k.idx[0]);
    IBGS_HIP(hipGetLastError());
    int rc = radix_sort_pairs(s, k.codes, k.idx, (size_t)P, 30, k.hist, k.hist_elems);
    if (rc) return rc;
    const unsigned nboxes = (unsigned)((P + KNN_BOX - 1) / KNN_BOX);
    hipLaunchKernelGGL(knn_boxes_kernel, dim3(nboxes), dim3(KNN_BOX), 0, s, (uint32_t)P, points, k.idx[0],
                       reinterpret_cast<float4*>(k.sorted), reinterpret_cast<float4*>(k.boxes));
    hipLaunchKernelGGL(knn_search_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, s, (uint32_t)P,
                       reinterpret_cast<const float4*>(k.sorted), reinterpret_cast<const float4*>(k.boxes), k.idx[0], out);
    IBGS_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
