// IBGS_FLAG_DETERMINISTIC: the render backward without float atomics (SURVEY 7.2 "alternative without atomics", 7.4 item 7).
//
// The reference (and the default mode here) accumulates per-Gaussian gradients with atomicAdd (backward.cu:673, 770, 793-804),
// so two runs differ in the last bits.  In this mode render_bwd.hip STORES the wave totals of every (Gaussian, tile) list entry
// in a slab row of its own -- row = (position of the entry in the sorted list) x (waves per tile) + (wave of the tile): the
// quadrant / half-tile variants run 4 / 2 waves per tile -- and this file sums the rows of each Gaussian in a fixed order:
//   1. (Gaussian id, list position) pairs are radix-sorted by id with the library's stable LSD sort, so the positions of one
//      Gaussian stay ascending (tile-major);
//   2. per-Gaussian segment starts by binary search in the sorted ids (one lane per Gaussian);
//   3. 16 lanes per Gaussian (one per grad_acc column) walk the segment and add the slab rows sequentially.
// Every output of ibgs_backward is then a pure function of its inputs.  HBM-bound: slab write + read (2 x 64 B per entry), three
// sort passes over 8 B per entry.
#include "common.h"

namespace ibgs {

static int id_bits(size_t P)
{
    int b = 1;
    while (((size_t)1 << b) < P && b < 32) b++;
    return b;
}

DetState DetState::carve(char* base, size_t R, size_t P, size_t* total)
{
    Carver c(base);
    DetState d;
    d.slab = c.take<float>(R * GACC_FLOATS);
    d.keys[0] = c.take<uint32_t>(R); d.keys[1] = c.take<uint32_t>(R);
    d.vals[0] = c.take<uint32_t>(R); d.vals[1] = c.take<uint32_t>(R);
    d.seg = c.take<uint32_t>(P + 1);
    d.hist_elems = radix_hist_elems(R);
    d.hist = c.take<uint32_t>(d.hist_elems);
    if (total) *total = (size_t)(c.cur - reinterpret_cast<uintptr_t>(base)) + 128;
    return d;
}

// (`listed`: the entries the forward's lists really hold, a device word -- fewer than R when a depth-bound hint shortened them (ibgs_rast.h: the forward
// returns R of the unbounded lists); the slots behind them get the id P, sort behind every Gaussian and belong to no segment)
__global__ void __launch_bounds__(256) det_pairs_kernel(const uint32_t* __restrict__ point_list, uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals, size_t rows, int ipt, const uint32_t* __restrict__ listed, uint32_t P)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < rows) { const size_t e = i / (size_t)ipt; keys[i] = e < (size_t)*listed ? point_list[e] : P; vals[i] = (uint32_t)i; }
}

// first sorted slot whose id is >= g, for g = 0..P (seg[P] = R)
__global__ void __launch_bounds__(256) det_segments_kernel(const uint32_t* __restrict__ sorted_ids, uint32_t* __restrict__ seg, size_t R, int P)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g > P) return;
    size_t lo = 0, hi = R;
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if (sorted_ids[mid] < (uint32_t)g) lo = mid + 1; else hi = mid;
    }
    seg[g] = (uint32_t)lo;
}

__global__ void __launch_bounds__(256) det_reduce_kernel(const float* __restrict__ slab, const uint32_t* __restrict__ pos,
                                                         const uint32_t* __restrict__ seg, float* __restrict__ gacc, int P)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int g = (int)(t >> 4), c = (int)(t & 15);
    if (g >= P) return;
    float acc = 0.f;
    for (uint32_t i = seg[g], e = seg[g + 1]; i < e; i++) acc += slab[(size_t)pos[i] * GACC_FLOATS + c];     // fixed order: ascending list position
    gacc[(size_t)g * GACC_FLOATS + c] = acc;
}

int launch_det_prepare(hipStream_t s, const DetState& d, size_t rows)
{
    IBGS_HIP(hipMemsetAsync(d.slab, 0, rows * GACC_FLOATS * sizeof(float), s));   // entries no wave reaches (behind saturation) stay zero
    return 0;
}

int launch_det_reduce(hipStream_t s, const DetState& d, const uint32_t* point_list, size_t R, int ipt, int P, float* gacc, const uint32_t* listed)
{
    const size_t rows = R * (size_t)ipt;
    hipLaunchKernelGGL(det_pairs_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, point_list, d.keys[0], d.vals[0], rows, ipt, listed, (uint32_t)P);
    IBGS_HIP(hipGetLastError());
    uint32_t* keys[2] = {d.keys[0], d.keys[1]};
    uint32_t* vals[2] = {d.vals[0], d.vals[1]};
    int rc = radix_sort_pairs(s, keys, vals, rows, id_bits((size_t)P + 1), d.hist, d.hist_elems);
    if (rc) return rc;
    hipLaunchKernelGGL(det_segments_kernel, dim3((unsigned)((P + 1 + 255) / 256)), dim3(256), 0, s, d.keys[0], d.seg, rows, P);
    IBGS_HIP(hipGetLastError());
    hipLaunchKernelGGL(det_reduce_kernel, dim3((unsigned)(((size_t)P * 16 + 255) / 256)), dim3(256), 0, s, d.slab, d.vals[0], d.seg, gacc, P);
    IBGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace ibgs
