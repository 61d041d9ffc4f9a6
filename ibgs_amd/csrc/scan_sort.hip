// A2 / A4: device-wide exclusive scan and stable LSD radix sort of (u32 key, u32 value) pairs.
//
// Replaces cub::DeviceScan::InclusiveSum and cub::DeviceRadixSort::SortPairs of the reference
// (DPR/cuda_rasterizer/rasterizer_impl.cu:426, 452-457).  The pipeline that uses them is different
// from the reference's (see binning.hip): Gaussians are depth-sorted FIRST (P items, 32-bit keys),
// duplicates are emitted in depth order, and only the tile id (<= 16 bits) is radix-sorted over the
// R duplicates -- two 7-bit passes at 1080p instead of six 8-bit passes over 64-bit keys.
//
// Kernels are written for wave64: ranking inside a pass uses 64-bit ballots ("match any") and
// per-wave LDS counters; scatter is staged through LDS so that global writes are runs of
// consecutive addresses.
#include "common.h"
#include <cstdlib>

namespace ibgs {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;   // 2048

constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 16;
constexpr int RS_CHUNK = RS_THREADS * RS_ITEMS;         // 4096
constexpr int RS_MAX_BINS = 256;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, WAVE);
        if (lane >= d) v += o;
    }
    return v;
}

// Exclusive scan of one value per thread across a 256-thread block. Returns the exclusive prefix;
// *total receives the block sum (valid in every thread).
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* total, uint32_t* lds_wave /*[4]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63 && wave < 4) lds_wave[wave] = inc;          // (callers with more than four waves pass zeros in the others)
    __syncthreads();
    const uint32_t w0 = lds_wave[0], w1 = lds_wave[1], w2 = lds_wave[2], w3 = lds_wave[3];
    uint32_t base = 0;
    if (wave > 0) base += w0;
    if (wave > 1) base += w1;
    if (wave > 2) base += w2;
    *total = w0 + w1 + w2 + w3;
    __syncthreads();
    return base + inc - v;
}

// Level kernel: local exclusive scan of a 2048-element chunk; block totals go to sums[block].
// When `single` (grid of one block) the grand total is written to out[n] if requested.
__global__ void __launch_bounds__(SCAN_THREADS) scan_chunk_kernel(const uint32_t* in, uint32_t* out,
                                                                  size_t n, uint32_t* __restrict__ sums, int write_total_single)
{
    __shared__ uint32_t lds_wave[4];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t local = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        local += v[k];
    }
    uint32_t total;
    uint32_t pre = block_exclusive_scan_256(local, &total, lds_wave);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = pre;
        pre += v[k];
    }
    if (threadIdx.x == 0) {
        if (sums) sums[blockIdx.x] = total;
        if (write_total_single) out[n] = total;
    }
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_add_kernel(uint32_t* __restrict__ out, size_t n, const uint32_t* __restrict__ sums,
                                                                size_t nblocks, int write_total)
{
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * SCAN_ITEMS;
    const uint32_t add = sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n) out[base + k] += add;
    if (write_total && blockIdx.x == 0 && threadIdx.x == 0) out[n] = sums[nblocks];
}

size_t scan_scratch_elems(size_t n)
{
    size_t total = 0;
    while (n > SCAN_CHUNK) {
        n = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
        total += n + 1 + 32;   // block sums + total slot (+pad)
    }
    return total + 64;
}

int exclusive_scan_u32(hipStream_t s, const uint32_t* in, uint32_t* out, size_t n, uint32_t* scratch,
                       size_t scratch_elems, bool with_total)
{
    if (n == 0) {
        if (with_total) IBGS_HIP(hipMemsetAsync(out, 0, sizeof(uint32_t), s));
        return 0;
    }
    const size_t nblocks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (nblocks == 1) {
        hipLaunchKernelGGL(scan_chunk_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, in, out, n, (uint32_t*)nullptr, with_total ? 1 : 0);
        IBGS_HIP(hipGetLastError());
        return 0;
    }
    if (scratch_elems < nblocks + 1) { set_error("scan scratch too small"); return -IBGS_ERR_ALLOC; }
    uint32_t* sums = scratch;
    hipLaunchKernelGGL(scan_chunk_kernel, dim3((unsigned)nblocks), dim3(SCAN_THREADS), 0, s, in, out, n, sums, 0);
    IBGS_HIP(hipGetLastError());
    const size_t used = nblocks + 1 + 32;
    int rc = exclusive_scan_u32(s, sums, sums, nblocks, scratch + used, scratch_elems > used ? scratch_elems - used : 0, true);
    if (rc) return rc;
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nblocks), dim3(SCAN_THREADS), 0, s, out, n, sums, nblocks, with_total ? 1 : 0);
    IBGS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// radix pass
// ------------------------------------------------------------------------------------------------
// (The classic hist + scan + scatter passes below stay generic over the key width and accept a device-side element count `n_dev`: round 1's
// R-sized tile sort used both; since the two-level binning of round 2 only the depth sort of more than 2 M Gaussians, the knn codes and
// the deterministic backward's ids come through here, all with 32-bit keys and a host-side count.)
// Rank of an element among the elements of its wave that carry the same digit, in lane order (what makes the pass
// stable), plus the running per-wave digit counter.  The set of lanes with the same digit ("match-any") comes from
// the LDS: every lane ORs its lane bit into the digit's 64-bit slot, then reads the slot back -- LDS operations of
// one wave execute in program order, so the read sees the whole wave's bits; the lowest lane of each group advances
// the counter and clears the slot for the next element.  3 LDS instructions instead of 7-8 ballot rounds (~50 VALU).
__device__ __forceinline__ uint32_t wave_rank(uint32_t d, bool valid, int lane, uint64_t lt_mask, uint32_t* wcnt_row, unsigned long long* peer_row)
{
    uint32_t r = 0;
    if (valid) {
        atomicOr(&peer_row[d], 1ull << lane);
        const unsigned long long peers = *reinterpret_cast<volatile unsigned long long*>(&peer_row[d]);
        volatile uint32_t* wc = wcnt_row;
        const uint32_t pre = wc[d];
        r = pre + (uint32_t)__popcll(peers & lt_mask);
        if ((peers & lt_mask) == 0ull) {            // lowest lane of the group
            wc[d] = pre + (uint32_t)__popcll(peers);
            *reinterpret_cast<volatile unsigned long long*>(&peer_row[d]) = 0ull;
        }
    }
    return r;
}

// K = uint32_t (uint16_t was round 1's tile sort of frames with <= 65536 tiles).
template <typename K>
__global__ void __launch_bounds__(RS_THREADS) radix_hist_kernel(const K* __restrict__ keys, size_t n, const uint32_t* __restrict__ n_dev,
                                                                int shift, int nbins, uint32_t* __restrict__ hist, unsigned nblocks)
{
    __shared__ uint32_t h[RS_MAX_BINS];
    if (n_dev) n = min(n, (size_t)*n_dev);
    for (int k = threadIdx.x; k < nbins; k += RS_THREADS) h[k] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_CHUNK;
    const uint32_t mask = (uint32_t)nbins - 1;
    // Each thread counts RS_ITEMS CONSECUTIVE keys (16-byte loads) and merges runs of equal digits before touching the LDS:
    // after the first pass neighbours mostly share the next digit, and 64 lanes adding 1 to the same LDS word serialise.
    const size_t i0 = base + (size_t)threadIdx.x * RS_ITEMS;
    uint32_t kv[RS_ITEMS];
    if (i0 + RS_ITEMS <= n) {
        constexpr int PER = 16 / (int)sizeof(K);                 // keys per 16-byte load
        const uint4* src = reinterpret_cast<const uint4*>(keys + i0);
#pragma unroll
        for (int v = 0; v < RS_ITEMS / PER; v++) {
            const uint4 q = src[v];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < PER; e++)
                kv[v * PER + e] = (sizeof(K) == 4) ? w[e] : ((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu);
        }
    } else {
#pragma unroll
        for (int k = 0; k < RS_ITEMS; k++) kv[k] = (i0 + k < n) ? (uint32_t)keys[i0 + k] : 0xFFFFFFFFu;
    }
    const int nvalid = (i0 >= n) ? 0 : (int)((n - i0 < (size_t)RS_ITEMS) ? (n - i0) : (size_t)RS_ITEMS);
    uint32_t run_d = 0, run_n = 0;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        if (k < nvalid) {
            const uint32_t d = (kv[k] >> shift) & mask;
            if (run_n != 0 && d != run_d) { atomicAdd(&h[run_d], run_n); run_n = 0; }
            run_d = d; run_n++;
        }
    }
    if (run_n != 0) atomicAdd(&h[run_d], run_n);
    __syncthreads();
    for (int k = threadIdx.x; k < nbins; k += RS_THREADS) hist[(size_t)k * nblocks + blockIdx.x] = h[k];
}

// Stable scatter of one 4096-element chunk. Element order inside the chunk is
// wave * 1024 + step * 64 + lane, i.e. each wave owns a contiguous quarter and walks it in
// 64-element steps, so (earlier wave, earlier step, lower lane) == earlier input position.
template <typename K>
__global__ void __launch_bounds__(RS_THREADS) radix_scatter_kernel(const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                   K* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                   size_t n, const uint32_t* __restrict__ n_dev, int shift, int nbits, int nbins,
                                                                   const uint32_t* __restrict__ hist_scanned, unsigned nblocks)
{
    __shared__ uint32_t wcnt[4][RS_MAX_BINS];
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * RS_CHUNK >= n) return;        // whole workgroup past the end (device-side count)
    __shared__ unsigned long long ptab[4][RS_MAX_BINS];      // match-any slots (wave_rank)
    __shared__ uint32_t dstart[RS_MAX_BINS];
    __shared__ uint32_t delta[RS_MAX_BINS];
    __shared__ uint32_t lds_wave[4];
    __shared__ uint32_t skey[RS_CHUNK];
    __shared__ uint32_t sval[RS_CHUNK];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < 4 * RS_MAX_BINS; k += RS_THREADS) { (&wcnt[0][0])[k] = 0; (&ptab[0][0])[k] = 0ull; }
    __syncthreads();

    const size_t base = (size_t)blockIdx.x * RS_CHUNK;
    const uint32_t mask = (uint32_t)nbins - 1;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const size_t idx = base + (size_t)wave * (RS_ITEMS * 64) + (size_t)k * 64 + lane;
        const bool valid = idx < n;
        key[k] = valid ? (uint32_t)keys_in[idx] : 0xFFFFFFFFu;
        val[k] = valid ? vals_in[idx] : 0u;
        const uint32_t d = (key[k] >> shift) & mask;
        const uint32_t r = wave_rank(d, valid, lane, lt_mask, &wcnt[wave][0], &ptab[wave][0]);
        rank[k] = r;
    }
    __syncthreads();

    // per digit: exclusive prefix over the 4 waves, block total, exclusive scan over digits
    uint32_t tot = 0;
    if (threadIdx.x < nbins) {
        const uint32_t c0 = wcnt[0][threadIdx.x], c1 = wcnt[1][threadIdx.x], c2 = wcnt[2][threadIdx.x], c3 = wcnt[3][threadIdx.x];
        wcnt[0][threadIdx.x] = 0; wcnt[1][threadIdx.x] = c0; wcnt[2][threadIdx.x] = c0 + c1; wcnt[3][threadIdx.x] = c0 + c1 + c2;
        tot = c0 + c1 + c2 + c3;
    }
    uint32_t blocktotal;
    const uint32_t ds = block_exclusive_scan_256(tot, &blocktotal, lds_wave);
    if (threadIdx.x < nbins) {
        dstart[threadIdx.x] = ds;
        delta[threadIdx.x] = hist_scanned[(size_t)threadIdx.x * nblocks + blockIdx.x] - ds;
    }
    __syncthreads();

    const size_t remaining = n - base;
    const uint32_t count = remaining < (size_t)RS_CHUNK ? (uint32_t)remaining : (uint32_t)RS_CHUNK;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const size_t idx = base + (size_t)wave * (RS_ITEMS * 64) + (size_t)k * 64 + lane;
        if (idx < n) {
            const uint32_t d = (key[k] >> shift) & mask;
            const uint32_t lpos = dstart[d] + wcnt[wave][d] + rank[k];
            skey[lpos] = key[k];
            sval[lpos] = val[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const uint32_t l = (uint32_t)k * RS_THREADS + threadIdx.x;
        if (l < count) {
            const uint32_t kk = skey[l];
            const uint32_t d = (kk >> shift) & mask;
            const uint32_t dst = l + delta[d];     // wraps correctly in uint32 arithmetic
            keys_out[dst] = (K)kk;
            vals_out[dst] = sval[l];
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Single-launch-per-pass variant ("onesweep" style): one kernel reads the keys once and builds the global
// digit histograms of ALL passes; each pass is then ONE kernel in which a workgroup ranks its chunk, publishes
// its per-digit counts and obtains its exclusive prefix from its predecessors by decoupled look-back
// (status word = 2 flag bits + 30-bit count, written / polled with agent-scope relaxed atomics = `sc1`
// stores and loads, MI355X_MICROARCH.md "Valid forms").  Chunks are handed out through an atomic ticket, so a
// workgroup only ever waits for workgroups that are already running -- no assumption on dispatch order or
// co-residency.  Replaces hist + 3 scan launches + scatter per pass by one launch per pass.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t OS_FLAG_AGG = 1u << 30, OS_FLAG_INC = 2u << 30, OS_VAL_MASK = (1u << 30) - 1u;
constexpr int OS_MAX_PASS = 4;
constexpr int OS_HIST_THREADS = 1024;      // few, large workgroups: every workgroup ends with one global atomic per non-empty bin
constexpr int OS_THREADS = 512, OS_WAVES = OS_THREADS / 64, OS_ITEMS = RS_CHUNK / OS_THREADS;      // same chunk as the classic passes, twice the waves: the ranking is a chain of LDS round trips per item

__global__ void __launch_bounds__(OS_HIST_THREADS) onesweep_hist_kernel(const uint32_t* __restrict__ keys, size_t n, int npass, int dbits,
                                                                   uint32_t* __restrict__ ghist /* npass x 256 */,
                                                                   int drop_max /* keys 0xFFFFFFFF take no part */, uint32_t* __restrict__ n_kept)
{
    __shared__ uint32_t h[OS_MAX_PASS][RS_MAX_BINS];
    __shared__ uint32_t s_kept;
    for (int k = threadIdx.x; k < OS_MAX_PASS * RS_MAX_BINS; k += OS_HIST_THREADS) (&h[0][0])[k] = 0;
    if (threadIdx.x == 0) s_kept = 0;
    __syncthreads();
    const uint32_t mask = (1u << dbits) - 1u;
    const size_t stride = (size_t)gridDim.x * OS_HIST_THREADS;
    uint32_t kept = 0;
    for (size_t i = (size_t)blockIdx.x * OS_HIST_THREADS + threadIdx.x; i < n; i += stride) {
        const uint32_t k = keys[i];
        if (drop_max && k == 0xFFFFFFFFu) continue;
        kept++;
        for (int p = 0; p < npass; p++) atomicAdd(&h[p][(k >> (p * dbits)) & mask], 1u);
    }
    if (kept) atomicAdd(&s_kept, kept);
    __syncthreads();
    if (threadIdx.x == 0 && s_kept) atomicAdd(n_kept, s_kept);
    for (int k = threadIdx.x; k < npass * RS_MAX_BINS; k += OS_HIST_THREADS) {
        const uint32_t c = (&h[0][0])[k];
        if (c) atomicAdd(&ghist[k], c);
    }
}

__global__ void __launch_bounds__(OS_THREADS) onesweep_pass_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                   uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                   size_t n, int shift, int nbits, int nbins,
                                                                   const uint32_t* __restrict__ ghist /* 256, this pass */,
                                                                   uint32_t* __restrict__ status /* nblocks x 256, zeroed */,
                                                                   uint32_t* __restrict__ ticket, uint32_t* __restrict__ err,
                                                                   int drop_here /* first pass of a sort that drops the 0xFFFFFFFF keys */,
                                                                   const uint32_t* __restrict__ n_kept /* items that take part (device) */,
                                                                   uint32_t* __restrict__ stays /* last pass only, or nullptr: a pass that would move nothing may leave its input where it is and say so */,
                                                                   uint32_t spin_limit /* look-back: sleeps on an unpublished word before the pass gives up (sets *err) */)
{
    __shared__ uint32_t wcnt[OS_WAVES][RS_MAX_BINS];
    __shared__ unsigned long long ptab[OS_WAVES][RS_MAX_BINS];      // match-any slots (wave_rank)
    __shared__ uint32_t dstart[RS_MAX_BINS];
    __shared__ uint32_t delta[RS_MAX_BINS];
    __shared__ uint32_t lds_wave[4];
    __shared__ uint32_t skey[RS_CHUNK];
    __shared__ uint32_t sval[RS_CHUNK];
    __shared__ uint32_t s_bid;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t kept = *n_kept;
    const size_t n_in = drop_here ? n : (size_t)kept;         // after the first pass the dropped keys are gone
    if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u);
    for (int k = threadIdx.x; k < OS_WAVES * RS_MAX_BINS; k += OS_THREADS) { (&wcnt[0][0])[k] = 0; (&ptab[0][0])[k] = 0ull; }
    __syncthreads();
    const uint32_t bid = s_bid;
    const size_t base = (size_t)bid * RS_CHUNK;
    if (base >= n_in) return;                               // (uniform; nobody looks back at a workgroup without items)

    // Every key carries the same digit (e.g. the top byte of depths within [2, 8)): the pass would not move anything.  The decision
    // comes from the global histogram, so every workgroup takes it alike; a pass that also has to drop keys always runs.
    const int all_one_bin = __syncthreads_or((int)(threadIdx.x < (unsigned)nbins && kept != 0u && ghist[threadIdx.x] == kept));
    if (all_one_bin && (!drop_here || (size_t)kept == n)) {
        if (stays) { if (bid == 0 && threadIdx.x == 0) *stays = 1u; return; }          // the caller reads the result from this pass's INPUT buffers
#pragma unroll
        for (int k = 0; k < OS_ITEMS; k++) {
            const size_t idx = base + (size_t)k * OS_THREADS + threadIdx.x;
            if (idx < n_in) { keys_out[idx] = keys_in[idx]; vals_out[idx] = vals_in[idx]; }
        }
        return;
    }

    const uint32_t mask = (uint32_t)nbins - 1;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    uint32_t key[OS_ITEMS], val[OS_ITEMS], rank[OS_ITEMS];
    uint32_t takes = 0;                                    // bit k: item k of this thread takes part
#pragma unroll
    for (int k = 0; k < OS_ITEMS; k++) {
        const size_t idx = base + (size_t)wave * (OS_ITEMS * 64) + (size_t)k * 64 + lane;
        const bool inside = idx < n_in;
        key[k] = inside ? keys_in[idx] : 0xFFFFFFFFu;
        val[k] = inside ? vals_in[idx] : 0u;
        const bool valid = inside && !(drop_here && key[k] == 0xFFFFFFFFu);
        takes |= valid ? (1u << k) : 0u;
        const uint32_t d = (key[k] >> shift) & mask;
        const uint32_t r = wave_rank(d, valid, lane, lt_mask, &wcnt[wave][0], &ptab[wave][0]);
        rank[k] = r;
    }
    __syncthreads();

    uint32_t tot = 0;
    if (threadIdx.x < nbins) {
#pragma unroll
        for (int w = 0; w < OS_WAVES; w++) { const uint32_t c = wcnt[w][threadIdx.x]; wcnt[w][threadIdx.x] = tot; tot += c; }          // exclusive over the waves
        // publish this chunk's count, then walk back over the predecessors
        uint32_t* mine = status + (size_t)bid * RS_MAX_BINS + threadIdx.x;
        __hip_atomic_store(mine, (bid == 0 ? OS_FLAG_INC : OS_FLAG_AGG) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // exclusive scan over digits of the GLOBAL histogram (every workgroup recomputes these 256 values) and of the local totals
    uint32_t gtotal, ltotal;
    const uint32_t gstart = block_exclusive_scan_256(threadIdx.x < nbins ? ghist[threadIdx.x] : 0u, &gtotal, lds_wave);
    const uint32_t ds = block_exclusive_scan_256(tot, &ltotal, lds_wave);
    if (spin_limit == 0u && bid == 0 && threadIdx.x == 0) *err = 1u;          // tests only (ibgs_debug_set_lookback_spins(0)): report a time-out that did not happen -- the sort itself is sound
    if (threadIdx.x < nbins) {
        uint32_t excl = 0;
        if (bid > 0) {
            // walk back over the predecessors LOOKBACK at a time: the status words of a batch are loaded together (independent loads, one
            // memory round trip), then consumed in order up to the first inclusive prefix -- or the first word not yet published, where
            // the walk resumes after a short sleep.  (One word per round trip made a pass of 150 workgroups that start together cost
            // 24 us: their aggregates are all there after ~4 us, the rest was walking.)
            constexpr int LOOKBACK = 8;
            int64_t b = (int64_t)bid - 1;
            uint32_t spins = 0;
            bool done = false;
            while (!done && b >= 0) {
                uint32_t st[LOOKBACK];
#pragma unroll
                for (int k = 0; k < LOOKBACK; k++)
                    st[k] = (b - k >= 0) ? __hip_atomic_load(status + (size_t)(b - k) * RS_MAX_BINS + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                         : OS_FLAG_INC;          // in front of the first workgroup: an inclusive prefix of 0
                int used = 0;
#pragma unroll
                for (int k = 0; k < LOOKBACK; k++) {
                    if (done || used != k) continue;               // stopped at an earlier word of this batch
                    const uint32_t flag = st[k] & ~OS_VAL_MASK;
                    if (flag == 0u) continue;                      // not published yet: resume here
                    excl += st[k] & OS_VAL_MASK;
                    used = k + 1;
                    if (flag == OS_FLAG_INC) done = true;
                }
                b -= used;
                if (!done && used < LOOKBACK) {                   // met an unpublished word
                    if (++spins > (spin_limit ? spin_limit : (1u << 26))) { *err = 1u; break; }      // bounded: never hang the device (0: the tests' hook above, with the default patience here)
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __hip_atomic_store(status + (size_t)bid * RS_MAX_BINS + threadIdx.x, OS_FLAG_INC | ((excl + tot) & OS_VAL_MASK),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        dstart[threadIdx.x] = ds;
        delta[threadIdx.x] = gstart + excl - ds;
    }
    __syncthreads();

    const uint32_t count = ltotal;                         // items of this chunk that take part
#pragma unroll
    for (int k = 0; k < OS_ITEMS; k++) {
        if (takes & (1u << k)) {
            const uint32_t d = (key[k] >> shift) & mask;
            const uint32_t lpos = dstart[d] + wcnt[wave][d] + rank[k];
            skey[lpos] = key[k];
            sval[lpos] = val[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < OS_ITEMS; k++) {
        const uint32_t l = (uint32_t)k * OS_THREADS + threadIdx.x;
        if (l < count) {
            const uint32_t kk = skey[l];
            const uint32_t d = (kk >> shift) & mask;
            const uint32_t dst = l + delta[d];
            keys_out[dst] = kk;
            vals_out[dst] = sval[l];
        }
    }
}

// Measured on MI355X (C3): depth sort (P = 1M, 245 chunks) 0.154 -> 0.135 ms, tile sort (R = 12.5M, 3052 chunks)
// 0.189 -> 0.254 ms: when ~1000 workgroups start together their look-back walks hundreds of AGGREGATE rows before
// the first INCLUSIVE prefix appears.  So it is used where it wins -- sorts of at most OS_AUTO_MAX_CHUNKS chunks (the
// launch-bound depth sort: 6 launches instead of 20) -- and the hist + scan + scatter passes everywhere else.
// IBGS_RADIX_ONESWEEP=1 / =0 forces it on / off for experiments.
static int g_use_onesweep = getenv("IBGS_RADIX_ONESWEEP") ? atoi(getenv("IBGS_RADIX_ONESWEEP")) : -1;   // -1 = by size (read once, when the library is loaded: tests/test_gpu_parity.py)
// look-back patience of the single-launch passes.  2^26 sleeps = seconds: only a starved workgroup ever gets there.  ibgs_debug_set_lookback_spins (tests): 0 makes
// every pass REPORT a time-out that did not happen (a real one cannot be staged: workgroups start in order, a predecessor has practically always published by the
// time its successor looks) -- how tests/test_gpu_async_sort_error.py drives the asynchronous error path; the sort itself stays sound
static uint32_t g_lookback_spins = 1u << 26;
void radix_set_onesweep(bool on) { g_use_onesweep = on ? 1 : 0; }
void radix_set_lookback_spins(uint32_t v) { g_lookback_spins = v; }
constexpr size_t OS_AUTO_MAX_CHUNKS = 4096;          // (round 1: 512 -- with 256-thread chunks the look-back of ~3000 workgroups lost against hist + scan + scatter; with 512-thread chunks and
                                                     // eight predecessors per look-back round trip the single-launch passes win up to at least 5 M keys: 0.178 vs 0.296 ms)

static size_t onesweep_elems(size_t n)
{
    const size_t nblocks = (n + RS_CHUNK - 1) / RS_CHUNK;
    return (size_t)OS_MAX_PASS * RS_MAX_BINS + 64 + (size_t)OS_MAX_PASS * (nblocks ? nblocks : 1) * RS_MAX_BINS;
}

static int radix_sort_pairs_onesweep(hipStream_t s, uint32_t* keys[2], uint32_t* vals[2], size_t n, int npass, int dbits,
                                     uint32_t* scratch, size_t scratch_elems, uint32_t* err_dev, uint32_t* kept_dev, bool scratch_is_zero, uint32_t* result_alt)
{
    const unsigned nblocks = (unsigned)((n + RS_CHUNK - 1) / RS_CHUNK);
    const int nbins = 1 << dbits;
    const size_t need = (size_t)OS_MAX_PASS * RS_MAX_BINS + 64 + (size_t)npass * nblocks * RS_MAX_BINS;
    if (scratch_elems < need) { set_error("onesweep scratch too small"); return -IBGS_ERR_ALLOC; }
    uint32_t* ghist = scratch;                                  // OS_MAX_PASS x 256
    uint32_t* tickets = scratch + OS_MAX_PASS * RS_MAX_BINS;      // per pass ticket counters, [32] = error flag, [33] = keys that take part
    uint32_t* status = tickets + 64;
    if (!scratch_is_zero) IBGS_HIP(hipMemsetAsync(scratch, 0, need * sizeof(uint32_t), s));
    const unsigned hb = (unsigned)((n + 8u * OS_HIST_THREADS - 1) / (8u * OS_HIST_THREADS));          // ~8 keys per thread (4 and 16 measured: 12.1 / 17.2 us against 12.2)
    const unsigned hblocks = hb < 256u ? (hb ? hb : 1u) : 256u;
    hipLaunchKernelGGL(onesweep_hist_kernel, dim3(hblocks), dim3(OS_HIST_THREADS), 0, s, keys[0], n, npass, dbits, ghist, kept_dev ? 1 : 0, kept_dev ? kept_dev : tickets + 33);
    IBGS_HIP(hipGetLastError());
    int cur = 0;
    for (int pass = 0; pass < npass; pass++) {
        hipLaunchKernelGGL(onesweep_pass_kernel, dim3(nblocks), dim3(OS_THREADS), 0, s, keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1],
                           n, pass * dbits, dbits, nbins, ghist + pass * RS_MAX_BINS, status + (size_t)pass * nblocks * RS_MAX_BINS,
                           tickets + pass, err_dev ? err_dev : tickets + 32, (kept_dev && pass == 0) ? 1 : 0, kept_dev ? kept_dev : tickets + 33,
                           (result_alt && npass == 4 && pass == 3) ? result_alt : (uint32_t*)nullptr, g_lookback_spins);
        IBGS_HIP(hipGetLastError());
        cur ^= 1;
    }
    if (cur != 0) {
        IBGS_HIP(hipMemcpyAsync(keys[0], keys[1], n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        IBGS_HIP(hipMemcpyAsync(vals[0], vals[1], n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    }
    if (!err_dev) {
        // A caller without an error word of its own (the deterministic backward's id sort, the knn's Morton sort: neither is on the
        // training step's fast path) gets the flag checked HERE: a look-back that gave up has scattered with a partial prefix, and a
        // mis-sorted result must fail the call instead of passing as "bit-identical" gradients.  One 4-byte read-back.
        uint32_t flag = 0;
        IBGS_HIP(hipMemcpyAsync(&flag, tickets + 32, sizeof(flag), hipMemcpyDeviceToHost, s));
        IBGS_HIP(hipStreamSynchronize(s));
        if (flag) { set_error("radix sort: a decoupled look-back timed out (result discarded)"); return -IBGS_ERR_HIP; }
    }
    return 0;
}

size_t radix_hist_elems(size_t n)
{
    const size_t nblocks = (n + RS_CHUNK - 1) / RS_CHUNK;
    const size_t hist = (size_t)RS_MAX_BINS * (nblocks ? nblocks : 1) + 1;
    const size_t classic = hist + 64 + scan_scratch_elems(hist);
    const size_t os = onesweep_elems(n);
    return classic > os ? classic : os;
}

__global__ void __launch_bounds__(RS_THREADS) count_kept_kernel(const uint32_t* __restrict__ keys, size_t n, uint32_t* __restrict__ kept)
{
    uint32_t c = 0;
    for (size_t i = (size_t)blockIdx.x * RS_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * RS_THREADS) c += keys[i] != 0xFFFFFFFFu ? 1u : 0u;
    for (int d = 32; d >= 1; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(kept, c);
}

// how many leading words of the scratch radix_sort_pairs wants zeroed (0: none); a caller that zeroes them itself -- e.g. inside a kernel
// it runs anyway -- passes scratch_is_zero and saves the fill launch
size_t radix_zero_elems(size_t n, int nbits_total)
{
    if (n == 0 || nbits_total <= 0) return 0;
    const size_t nblocks = (n + RS_CHUNK - 1) / RS_CHUNK;
    const int npass = (nbits_total + 7) / 8;
    const bool want_os = g_use_onesweep >= 0 ? g_use_onesweep != 0 : nblocks <= OS_AUTO_MAX_CHUNKS;
    if (!(want_os && npass <= OS_MAX_PASS && n < (size_t)OS_VAL_MASK)) return 0;
    return (size_t)OS_MAX_PASS * RS_MAX_BINS + 64 + (size_t)npass * nblocks * RS_MAX_BINS;
}

int radix_sort_pairs(hipStream_t s, uint32_t* keys[2], uint32_t* vals[2], size_t n, int nbits_total,
                     uint32_t* hist, size_t hist_elems, uint32_t* err_dev, uint32_t* kept_dev, bool scratch_is_zero, uint32_t* result_alt)
{
    if (n == 0 || nbits_total <= 0) return 0;
    const unsigned nblocks = (unsigned)((n + RS_CHUNK - 1) / RS_CHUNK);
    const int npass = (nbits_total + 7) / 8;
    const int dbits = (nbits_total + npass - 1) / npass;
    const int nbins = 1 << dbits;
    const bool want_os = g_use_onesweep >= 0 ? g_use_onesweep != 0 : nblocks <= OS_AUTO_MAX_CHUNKS;
    if (want_os && npass <= OS_MAX_PASS && n < (size_t)OS_VAL_MASK)
        return radix_sort_pairs_onesweep(s, keys, vals, n, npass, dbits, hist, hist_elems, err_dev, kept_dev, scratch_is_zero, result_alt);
    const size_t hist_n = (size_t)nbins * nblocks;
    if (hist_elems < hist_n + 1 + 64) { set_error("radix scratch too small"); return -IBGS_ERR_ALLOC; }
    uint32_t* scan_scratch = hist + hist_n + 1 + 63;
    const size_t scan_elems = hist_elems - (hist_n + 1 + 63);
    if (kept_dev) {      // the 0xFFFFFFFF keys sort behind everything else: counting the others is all that "dropping" them takes here
        hipLaunchKernelGGL(count_kept_kernel, dim3(nblocks < 1024u ? nblocks : 1024u), dim3(RS_THREADS), 0, s, keys[0], n, kept_dev);
        IBGS_HIP(hipGetLastError());
    }
    int cur = 0;
    for (int pass = 0; pass < npass; pass++) {
        const int shift = pass * dbits;
        hipLaunchKernelGGL(radix_hist_kernel<uint32_t>, dim3(nblocks), dim3(RS_THREADS), 0, s, keys[cur], n, (const uint32_t*)nullptr, shift, nbins, hist, nblocks);
        IBGS_HIP(hipGetLastError());
        int rc = exclusive_scan_u32(s, hist, hist, hist_n, scan_scratch, scan_elems, false);
        if (rc) return rc;
        hipLaunchKernelGGL(radix_scatter_kernel<uint32_t>, dim3(nblocks), dim3(RS_THREADS), 0, s, keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1],
                           n, (const uint32_t*)nullptr, shift, dbits, nbins, hist, nblocks);
        IBGS_HIP(hipGetLastError());
        cur ^= 1;
    }
    if (cur != 0) {   // odd number of passes: bring the result back to buffer 0
        IBGS_HIP(hipMemcpyAsync(keys[0], keys[1], n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        IBGS_HIP(hipMemcpyAsync(vals[0], vals[1], n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

}  // namespace ibgs
