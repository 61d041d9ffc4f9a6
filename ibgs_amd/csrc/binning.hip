// A3 / A5: duplicate emission in depth order and tile ranges.
//
// The reference emits (tile<<32 | depth) keys in Gaussian-index order, one thread per Gaussian with a
// divergent loop over its tile rectangle, and lets a 6-pass 64-bit radix sort establish both the tile
// and the depth order (DPR/cuda_rasterizer/rasterizer_impl.cu:187-228, 449-457).  Here the P
// Gaussians are already depth-sorted (stable, ties keep index order), so a duplicate only needs
// its tile id as key and the R-sized sort is a stable sort on <= 16 bits.  The resulting per-tile
// lists are identical to the reference's: ascending depth bits, ties in Gaussian-index order
// (SURVEY.md Q9).
//
// Emission is load-balanced over OUTPUT slots (a block owns 2048 consecutive slots and finds the
// Gaussians that cover them), so global writes are fully coalesced regardless of how large a
// Gaussian's rectangle is.
#include "common.h"

namespace ibgs {

constexpr int EM_THREADS = 256;
constexpr int EM_ITEMS = 8;
constexpr int EM_CHUNK = EM_THREADS * EM_ITEMS;   // 2048 output slots per block

__global__ void __launch_bounds__(256) gather_tiles_kernel(int P, const uint32_t* __restrict__ order,
                                                           const uint32_t* __restrict__ tiles, uint32_t* __restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < P) out[j] = tiles[order[j]];
}

int launch_gather_tiles(hipStream_t s, int P, const GeomState& g)
{
    hipLaunchKernelGGL(gather_tiles_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, g.sort_val[0], g.tiles, g.offsets);
    IBGS_HIP(hipGetLastError());
    return 0;
}

// Largest j in [0, n] with offs[j] <= target, found by one wave with a 64-ary search
// (offs is non-decreasing, offs[0] == 0, offs has n+1 entries).
__device__ __forceinline__ uint32_t wave_search_le(const uint32_t* __restrict__ offs, uint32_t n, uint32_t target, int lane)
{
    uint32_t lo = 0, len = n + 1;          // candidate indices [lo, lo+len)
    while (len > 1) {
        const uint32_t step = (len + 63) / 64;
        const uint32_t idx = lo + (uint32_t)lane * step;
        const bool ok = (idx < lo + len) && (offs[idx] <= target);
        const uint64_t bal = __ballot(ok);
        const int cnt = __popcll(bal);     // ok lanes form a prefix because offs is monotone
        const uint32_t nlo = lo + (uint32_t)(cnt - 1) * step;
        const uint32_t nend = min(lo + len, nlo + step);
        lo = nlo; len = nend - nlo;
    }
    return lo;
}

template <typename K>
__global__ void __launch_bounds__(EM_THREADS) emit_kernel(uint32_t P, uint32_t R, const uint32_t* __restrict__ R_dev, int gx,
                                                          const uint32_t* __restrict__ order,
                                                          const uint32_t* __restrict__ offs /* P+1 */,
                                                          const uint32_t* __restrict__ rect,
                                                          const uint64_t* __restrict__ tmask,
                                                          K* __restrict__ keys, uint32_t* __restrict__ vals)
{
    __shared__ uint32_t win[EM_CHUNK + 2];
    __shared__ uint32_t jrange[2];
    if (R_dev) R = min(R, *R_dev);          // launch sized for an upper bound, real count on the device (api.hip)
    const uint32_t s0 = blockIdx.x * (uint32_t)EM_CHUNK;
    if (s0 >= R) return;
    const uint32_t s1 = min(R, s0 + (uint32_t)EM_CHUNK);   // exclusive
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        const uint32_t j = wave_search_le(offs, P, s0, lane);
        if (lane == 0) jrange[0] = j;
    } else if (wave == 1) {
        const uint32_t j = wave_search_le(offs, P, s1 - 1, lane);
        if (lane == 0) jrange[1] = j;
    }
    __syncthreads();
    // Several Gaussians with zero tiles can share offs[j] == target; the search returns the LAST
    // index with offs <= target, which is the one that actually owns the slot (its successor
    // starts strictly later).
    const uint32_t j_lo = jrange[0], j_hi = jrange[1];
    const uint32_t nwin = j_hi - j_lo + 2;                  // offs[j_lo .. j_hi+1]
    for (uint32_t k = threadIdx.x; k < nwin; k += EM_THREADS) win[k] = offs[j_lo + k];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < EM_ITEMS; it++) {
        const uint32_t s = s0 + (uint32_t)it * EM_THREADS + threadIdx.x;
        if (s < s1) {
            // binary search in the window: largest w in [0, nwin-2] with win[w] <= s
            uint32_t lo = 0, hi = nwin - 1;                  // invariant: win[lo] <= s < win[hi] (win[nwin-1] > s)
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (win[mid] <= s) lo = mid; else hi = mid;
            }
            const uint32_t id = order[j_lo + lo];
            uint32_t k = s - win[lo];
            const uint32_t rx = rect[2 * id], ry = rect[2 * id + 1];
            const uint32_t x0 = rx & 0xFFFFu, x1 = rx >> 16, y0 = ry & 0xFFFFu, y1 = ry >> 16;
            const uint32_t w = x1 - x0;
            if (w * (y1 - y0) <= (uint32_t)IBGS_CULL_MAX_TILES) {
                // k-th surviving tile = position of the k-th set bit of the cull mask (up to four 64-bit words)
                const uint64_t* mw = tmask + (size_t)id * IBGS_CULL_WORDS;
                uint64_t m = mw[0];
                uint32_t pos = 0;
                if (w * (y1 - y0) > 64u) {
#pragma unroll
                    for (int r = 0; r < IBGS_CULL_WORDS - 1; r++) {
                        const uint32_t c = (uint32_t)__popcll(m);
                        if (k >= c) { k -= c; pos += 64u; m = mw[r + 1]; }
                        else break;
                    }
                }
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) {
                    const uint64_t low = m & ((1ull << sft) - 1ull);
                    const uint32_t c = (uint32_t)__popcll(low);
                    if (k >= c) { k -= c; m >>= sft; pos += (uint32_t)sft; } else { m = low; }
                }
                k = pos;
            }
            // k / w and k % w without an integer division: (k + 0.5) / w is at least 0.5 / w away from an integer, far more
            // than the error of the hardware reciprocal (k < 2^17, w <= 512)
            const uint32_t row = (uint32_t)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)w));
            const uint32_t ty = y0 + row, tx = x0 + (k - row * w);
            keys[s] = (K)(ty * (uint32_t)gx + tx);
            vals[s] = id;
        }
    }
}

int launch_emit(hipStream_t s, int P, int64_t R, int gx, const GeomState& g, const BinState& b, const uint32_t* R_dev, bool key16)
{
    if (R <= 0) return 0;
    const unsigned nblocks = (unsigned)((R + EM_CHUNK - 1) / EM_CHUNK);
    if (key16) hipLaunchKernelGGL(emit_kernel<uint16_t>, dim3(nblocks), dim3(EM_THREADS), 0, s, (uint32_t)P, (uint32_t)R, R_dev, gx,
                                  g.sort_val[0], g.offsets, g.rect, g.tmask, reinterpret_cast<uint16_t*>(b.keys[0]), b.vals[0]);
    else hipLaunchKernelGGL(emit_kernel<uint32_t>, dim3(nblocks), dim3(EM_THREADS), 0, s, (uint32_t)P, (uint32_t)R, R_dev, gx,
                            g.sort_val[0], g.offsets, g.rect, g.tmask, b.keys[0], b.vals[0]);
    IBGS_HIP(hipGetLastError());
    return 0;
}

// identifyTileRanges, rasterizer_impl.cu:233-255 (ranges pre-zeroed by the caller)
template <typename K>
__global__ void __launch_bounds__(256) ranges_kernel(uint32_t R, const uint32_t* __restrict__ R_dev, const K* __restrict__ keys, uint32_t* __restrict__ ranges)
{
    if (R_dev) R = min(R, *R_dev);
    constexpr int N = 8;                                        // consecutive keys per thread (16-byte loads)
    const uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * (uint32_t)N;
    if (i0 >= R) return;
    uint32_t kv[N];
    if (i0 + N <= R) {
        constexpr int PER = 16 / (int)sizeof(K);
        const uint4* src = reinterpret_cast<const uint4*>(keys + i0);
#pragma unroll
        for (int v = 0; v < N / PER; v++) {
            const uint4 q = src[v];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < PER; e++) kv[v * PER + e] = (sizeof(K) == 4) ? w[e] : ((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; k++) kv[k] = (i0 + k < R) ? (uint32_t)keys[i0 + k] : 0u;
    }
    uint32_t prev = (i0 == 0) ? 0xFFFFFFFFu : (uint32_t)keys[i0 - 1];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const uint32_t i = i0 + k;
        if (i < R) {
            const uint32_t cur = kv[k];
            if (i == 0) ranges[2 * cur] = 0;
            else if (cur != prev) { ranges[2 * prev + 1] = i; ranges[2 * cur] = i; }
            if (i == R - 1) ranges[2 * cur + 1] = R;
            prev = cur;
        }
    }
}

int launch_ranges(hipStream_t s, int64_t R, int ntiles, const uint32_t* sorted_keys, uint32_t* ranges, const uint32_t* R_dev, bool key16)
{
    IBGS_HIP(hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)ntiles, s));
    if (R <= 0) return 0;
    if (key16) hipLaunchKernelGGL(ranges_kernel<uint16_t>, dim3((unsigned)((R + 2047) / 2048)), dim3(256), 0, s, (uint32_t)R, R_dev, reinterpret_cast<const uint16_t*>(sorted_keys), ranges);
    else hipLaunchKernelGGL(ranges_kernel<uint32_t>, dim3((unsigned)((R + 2047) / 2048)), dim3(256), 0, s, (uint32_t)R, R_dev, sorted_keys, ranges);
    IBGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace ibgs
