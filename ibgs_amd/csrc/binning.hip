// A3-A5: per-tile lists (duplicate emission, tile sort, tile ranges) by TWO-LEVEL binning.
//
// The reference emits one 64-bit (tile, depth) key per (Gaussian, tile) pair and sorts all R of them with a 6-pass radix sort
// (DPR/cuda_rasterizer/rasterizer_impl.cu:187-228, 449-457).  Round 1 here sorted the P Gaussians by depth first and the R
// duplicates by tile id only (2 x 7-bit passes over R).  This round the R-sized sort is gone:
//
//   1. coarse cells of 8 x 8 tiles: every Gaussian (in depth order) emits ONE entry per cell that holds at least one of its
//      surviving tiles -- (cell id, entry index) plus the 64-bit mask of its tiles inside the cell.  C3: 2.6 M entries
//      instead of 12.4 M;
//   2. the coarse entries are stably sorted by cell id (one 8-bit pass for a 1080p frame): every cell now holds its Gaussians
//      in depth order;
//   3. each cell is cut into chunks of 256 entries (one wave each).  A first kernel counts, per chunk, how many entries touch
//      each of the cell's 64 tiles; one wave per cell turns the counts into prefixes within the cell and per-tile totals; a
//      scan over the tiles (in global row-major order) gives every tile's range -- `ranges` falls out, no key array, no
//      range-finding pass;
//   4. a second kernel walks each chunk in order and writes the Gaussian ids to their final slots: for tile t the lanes whose
//      mask has bit t form a ballot; slot = start of the tile + prefix of the chunk + number of earlier lanes in the ballot.
//      Entry order inside a tile = order inside the cell = depth order, ties by Gaussian index.
//
// The lists are bit-identical to the ones a stable sort of (tile, depth-rank) keys produces (= the reference's lists, restricted
// to the tiles the exact cull keeps); tests compare them with the oracle entry by entry.
#include "common.h"

namespace ibgs {

constexpr int CB = BIN_CELL;            // tiles per cell edge (8)
constexpr int XCHUNK = BIN_XCHUNK;      // coarse entries per expansion chunk = 4 rounds of one wave

// up to 8 bits of a <= 256-bit row-major tile mask, starting at bit `start`
__device__ __forceinline__ uint32_t mask_bits(const uint64_t* __restrict__ mw, uint32_t start, uint32_t len)
{
    const uint32_t w = start >> 6, o = start & 63u;
    uint64_t v = mw[w] >> o;
    if (o + len > 64u) v |= mw[w + 1] << (64u - o);        // only reached for masks of more than one word
    return (uint32_t)v & ((1u << len) - 1u);
}

struct RectU { uint32_t x0, x1, y0, y1; };
__device__ __forceinline__ RectU load_rect(const uint32_t* __restrict__ rect, uint32_t id)
{
    const uint32_t rx = rect[2 * id], ry = rect[2 * id + 1];
    return RectU{rx & 0xFFFFu, rx >> 16, ry & 0xFFFFu, ry >> 16};
}

// surviving tiles of one Gaussian inside cell (ccx, ccy): bit ly * 8 + lx for tile (8 ccx + lx, 8 ccy + ly)
__device__ __forceinline__ uint64_t cell_mask(const RectU& r, const uint64_t* __restrict__ mw, uint32_t ccx, uint32_t ccy)
{
    const uint32_t cx0 = ccx * CB, cy0 = ccy * CB;
    const uint32_t xa = max(r.x0, cx0), xb = min(r.x1, cx0 + CB), ya = max(r.y0, cy0), yb = min(r.y1, cy0 + CB);
    if (xa >= xb || ya >= yb) return 0ull;
    const uint32_t w = r.x1 - r.x0;
    const bool masked = w * (r.y1 - r.y0) <= (uint32_t)IBGS_CULL_MAX_TILES;       // larger rectangles keep every tile (preprocess.hip)
    const uint32_t len = xb - xa;
    uint64_t m = 0ull;
    for (uint32_t ty = ya; ty < yb; ty++) {
        const uint32_t bits = masked ? mask_bits(mw, (ty - r.y0) * w + (xa - r.x0), len) : ((1u << len) - 1u);
        m |= (uint64_t)bits << ((ty - cy0) * CB + (xa - cx0));
    }
    return m;
}

// Per depth rank j: tiles touched (for R, only when the host needs it before the binning) and the number of coarse slots the
// Gaussian gets: min(cells of its rectangle, surviving tiles) >= its non-empty cells, so C <= R always holds (arena sizing).
__global__ void __launch_bounds__(256) gather_tiles_kernel(int P, const uint32_t* __restrict__ order, const uint32_t* __restrict__ tc,
                                                           uint32_t* __restrict__ out_tiles /* may be null */, uint32_t* __restrict__ out_cells)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P) return;
    const uint32_t v = tc[order[j]];
    const uint32_t nt = v & 0x3FFFFu, nc = v >> 18;
    if (out_tiles) out_tiles[j] = nt;
    out_cells[j] = min(nt, nc);
}

int launch_gather_tiles(hipStream_t s, int P, const GeomState& g, bool want_tiles)
{
    hipLaunchKernelGGL(gather_tiles_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, g.sort_val[0], g.tc, want_tiles ? g.offsets : nullptr, g.coffs);
    IBGS_HIP(hipGetLastError());
    return 0;
}

// Four lanes per depth rank: lane c looks at cells c, c + 4, ... of the Gaussian's rectangle (row-major); the cells that hold a
// surviving tile are packed into the Gaussian's slots [coffs[j], coffs[j + 1]) (rank inside the group of four from a ballot), the
// slots that stay free become NULL entries (key = ncells: sorted behind every real cell and ignored).
__global__ void __launch_bounds__(256) coarse_emit_kernel(int P, uint32_t ccap, int cgx, uint32_t null_key, const uint32_t* __restrict__ order,
                                                          const uint32_t* __restrict__ coffs, const uint32_t* __restrict__ rect,
                                                          const uint64_t* __restrict__ tmask,
                                                          uint32_t* __restrict__ ckeys, uint32_t* __restrict__ cvals,
                                                          uint32_t* __restrict__ cid, uint64_t* __restrict__ cmask)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = min(gid >> 2, P - 1);               // every lane stays in the ballots below
    const bool live = (gid >> 2) < P;
    const uint32_t e0 = coffs[j], e1 = coffs[j + 1];
    // e1 > ccap: the arena was carved for a too small hint and the call is redone (api.hip); until then every slot below the
    // capacity must still hold a well-formed (null) entry, the sort reads them all
    const bool fits = e1 <= ccap;
    const bool work = live && e1 != e0 && fits;
    const uint32_t id = order[j];
    const RectU r = load_rect(rect, id);
    const uint64_t* mw = tmask + (size_t)id * IBGS_CULL_WORDS;
    const uint32_t c0x = r.x0 / CB, c0y = r.y0 / CB, ncx = work ? (r.x1 - 1) / CB - c0x + 1 : 1u;
    const uint32_t ncell = work ? ncx * ((r.y1 - 1) / CB - c0y + 1) : 0u;
    const int c = gid & 3, qshift = (int)(threadIdx.x & 63) & ~3;
    uint32_t filled = 0;
    // the four lanes of a Gaussian run the same number of rounds; the wave runs the maximum
    uint32_t nr = (ncell + 3) / 4;
    for (int d = 32; d >= 1; d >>= 1) nr = max(nr, (uint32_t)__shfl_xor((int)nr, d, 64));
    for (uint32_t rd = 0; rd < nr; rd++) {
        const uint32_t k = rd * 4 + (uint32_t)c;
        uint64_t m = 0ull; uint32_t cell = 0;
        if (k < ncell) {
            const uint32_t row = k / ncx, cx = c0x + (k - row * ncx), cy = c0y + row;
            m = cell_mask(r, mw, cx, cy);
            cell = cy * (uint32_t)cgx + cx;
        }
        const uint32_t qb = (uint32_t)(__builtin_amdgcn_ballot_w64(m != 0ull) >> qshift) & 0xFu;
        if (m != 0ull) {
            const uint32_t e = e0 + filled + (uint32_t)__popc(qb & ((1u << c) - 1u));
            ckeys[e] = cell; cvals[e] = e; cid[e] = id; cmask[e] = m;
        }
        filled += (uint32_t)__popc(qb);
    }
    if (live)
        for (uint32_t e = e0 + filled + (uint32_t)c; e < min(e1, ccap); e += 4) { ckeys[e] = null_key; cvals[e] = e; cid[e] = id; cmask[e] = 0ull; }
}

// First sorted entry of every cell + chunk bookkeeping, ONE workgroup (ncells <= a few thousand).  After a ONE-pass sort the
// scanned radix histogram already holds the answer: hist[digit * nblocks] = number of keys with a smaller digit.
__global__ void __launch_bounds__(256) cell_setup_kernel(const uint32_t* __restrict__ C_dev, uint32_t ccap, const uint32_t* __restrict__ sorted_cells,
                                                         const uint32_t* __restrict__ hist, unsigned hist_stride /* 0: search instead */,
                                                         int ncells, uint32_t* __restrict__ cell_start /* ncells + 1 */,
                                                         uint32_t* __restrict__ cell_chunk0 /* ncells + 1 */)
{
    __shared__ uint32_t s_part[256];
    const uint32_t C = min(*C_dev, ccap);
    for (int c = threadIdx.x; c <= ncells; c += 256) {
        uint32_t lo = 0;
        if (hist_stride) lo = min(hist[(size_t)c * hist_stride], C);
        else {      // first sorted entry with cell id >= c (the list is sorted by cell id, null entries last)
            uint32_t hi = C;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sorted_cells[mid] < (uint32_t)c) lo = mid + 1; else hi = mid; }
        }
        cell_start[c] = lo;
    }
    __syncthreads();
    // exclusive scan of chunks per cell (sequential per thread over a strip, then over the 256 strip sums)
    const int per = (ncells + 255) / 256;
    const int c0 = min(ncells, (int)threadIdx.x * per), c1 = min(ncells, c0 + per);
    uint32_t sum = 0;
    for (int c = c0; c < c1; c++) sum += (cell_start[c + 1] - cell_start[c] + XCHUNK - 1) / XCHUNK;
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int t = 0; t < 256; t++) { const uint32_t v = s_part[t]; s_part[t] = run; run += v; } cell_chunk0[ncells] = run; }
    __syncthreads();
    uint32_t run = s_part[threadIdx.x];
    for (int c = c0; c < c1; c++) { cell_chunk0[c] = run; run += (cell_start[c + 1] - cell_start[c] + XCHUNK - 1) / XCHUNK; }
}

// which cell owns chunk `ch` (cell_chunk0 is non-decreasing, cell_chunk0[ncells] = number of chunks)
__device__ __forceinline__ int cell_of_chunk(const uint32_t* __restrict__ cell_chunk0, int ncells, uint32_t ch)
{
    int lo = 0, hi = ncells;          // largest c with cell_chunk0[c] <= ch and cell_chunk0[c + 1] > ch
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cell_chunk0[mid] <= ch) lo = mid; else hi = mid; }
    return lo;
}

// One wave per chunk: counts of entries touching each of the cell's 64 tiles (lane = tile).  The chunk's four rounds of masks sit
// in registers; for each tile the four ballots are counted on the scalar unit and the sum lands in lane t with one v_writelane.
__global__ void __launch_bounds__(64) expand_count_kernel(const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ cell_chunk0, int ncells,
                                                          const uint32_t* __restrict__ sorted_e, const uint64_t* __restrict__ cmask,
                                                          uint32_t* __restrict__ chunk_cnt /* nchunks x 64 */)
{
    const uint32_t ch = blockIdx.x;
    if (ch >= cell_chunk0[ncells]) return;
    const int lane = threadIdx.x;
    const int cell = cell_of_chunk(cell_chunk0, ncells, ch);
    const uint32_t i0 = cell_start[cell] + (ch - cell_chunk0[cell]) * XCHUNK, i1 = min(cell_start[cell + 1], i0 + XCHUNK);
    uint32_t mlo[XCHUNK / 64], mhi[XCHUNK / 64];
#pragma unroll
    for (int rd = 0; rd < XCHUNK / 64; rd++) {
        const uint32_t i = i0 + (uint32_t)rd * 64 + lane;
        const uint64_t m = (i < i1) ? cmask[sorted_e[i]] : 0ull;
        mlo[rd] = (uint32_t)m; mhi[rd] = (uint32_t)(m >> 32);
    }
    int cnt = 0;                       // lane t: entries of this chunk whose mask has bit t
#pragma unroll
    for (int t = 0; t < 64; t++) {
        int c = 0;
#pragma unroll
        for (int rd = 0; rd < XCHUNK / 64; rd++)
            c += __popcll(__builtin_amdgcn_ballot_w64(((t < 32 ? mlo[rd] : mhi[rd]) & (1u << (t & 31))) != 0u));
        // v_writelane_b32: lane t of cnt <- c (uniform).  Inline asm: this hipcc has no builtin for it; one instruction, no hazard inside
        asm("v_writelane_b32 %0, %1, %2" : "+v"(cnt) : "s"(__builtin_amdgcn_readfirstlane(c)), "n"(t));
    }
    chunk_cnt[(size_t)ch * 64 + lane] = (uint32_t)cnt;
}

// One wave per cell (lane = tile of the cell): counts -> exclusive prefixes over the cell's chunks, per-tile totals
__global__ void __launch_bounds__(64) cell_scan_kernel(const uint32_t* __restrict__ cell_chunk0, int ncells, int cgx, int gx, int gy,
                                                       uint32_t* __restrict__ chunk_cnt, uint32_t* __restrict__ tile_total /* ntiles */)
{
    const int cell = blockIdx.x, lane = threadIdx.x;
    uint32_t run = 0;
    const uint32_t e = cell_chunk0[cell + 1];
    for (uint32_t ch = cell_chunk0[cell]; ch < e; ch += 8) {          // eight independent loads in flight, then the serial prefix
        uint32_t c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = (ch + k < e) ? chunk_cnt[(size_t)(ch + k) * 64 + lane] : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) if (ch + k < e) { chunk_cnt[(size_t)(ch + k) * 64 + lane] = run; run += c[k]; }
    }
    const int tx = (cell % cgx) * CB + (lane & 7), ty = (cell / cgx) * CB + (lane >> 3);
    if (tx < gx && ty < gy) tile_total[ty * gx + tx] = run;
}

// tile_start (exclusive scan of the totals, in place) -> ranges; empty tiles keep (0, 0) like identifyTileRanges
// (rasterizer_impl.cu:233-255 after its memset)
__global__ void __launch_bounds__(256) write_ranges_kernel(int ntiles, const uint32_t* __restrict__ tile_start /* ntiles + 1 */, uint32_t* __restrict__ ranges,
                                                           const uint32_t* __restrict__ C_dev, uint32_t* __restrict__ counters /* R, -, C */, uint32_t cap)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) { counters[0] = tile_start[ntiles]; counters[2] = *C_dev; }      // what the host reads back in ONE copy (api.hip)
    if (t >= ntiles) return;
    // clamped to the capacity of point_list: after a too small hint the render kernels must not walk past it (the call is redone)
    const uint32_t a = min(tile_start[t], cap), b = min(tile_start[t + 1], cap);
    ranges[2 * t] = (b > a) ? a : 0u; ranges[2 * t + 1] = (b > a) ? b : 0u;
}

// One wave per chunk: ids to their final slots, in order.  Tile by tile: the lanes whose mask has the bit form a ballot per round;
// slot = first slot of the tile for this chunk (scalar, from lane t) + entries of earlier rounds + earlier lanes of the ballot.
__global__ void __launch_bounds__(64) expand_scatter_kernel(const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ cell_chunk0, int ncells,
                                                            int cgx, int gx, int gy, const uint32_t* __restrict__ sorted_e,
                                                            const uint32_t* __restrict__ cid, const uint64_t* __restrict__ cmask,
                                                            const uint32_t* __restrict__ chunk_pref, const uint32_t* __restrict__ tile_start,
                                                            uint32_t cap, uint32_t* __restrict__ point_list)
{
    const uint32_t ch = blockIdx.x;
    if (ch >= cell_chunk0[ncells]) return;
    const int lane = threadIdx.x;
    const int cell = cell_of_chunk(cell_chunk0, ncells, ch);
    const uint32_t i0 = cell_start[cell] + (ch - cell_chunk0[cell]) * XCHUNK, i1 = min(cell_start[cell + 1], i0 + XCHUNK);
    // lane t: first slot of tile t for this chunk
    const int tx = (cell % cgx) * CB + (lane & 7), ty = (cell / cgx) * CB + (lane >> 3);
    const int slot = (tx < gx && ty < gy) ? (int)(tile_start[ty * gx + tx] + chunk_pref[(size_t)ch * 64 + lane]) : 0;
    uint32_t mlo[XCHUNK / 64], mhi[XCHUNK / 64], id[XCHUNK / 64];
#pragma unroll
    for (int rd = 0; rd < XCHUNK / 64; rd++) {
        const uint32_t i = i0 + (uint32_t)rd * 64 + lane;
        uint64_t m = 0ull; id[rd] = 0u;
        if (i < i1) { const uint32_t e = sorted_e[i]; m = cmask[e]; id[rd] = cid[e]; }
        mlo[rd] = (uint32_t)m; mhi[rd] = (uint32_t)(m >> 32);
    }
#pragma unroll
    for (int t = 0; t < 64; t++) {
        uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane(slot, t);
#pragma unroll
        for (int rd = 0; rd < XCHUNK / 64; rd++) {
            const bool mine = ((t < 32 ? mlo[rd] : mhi[rd]) & (1u << (t & 31))) != 0u;
            const uint64_t b = __builtin_amdgcn_ballot_w64(mine);
            const uint32_t n = (uint32_t)__popcll(b);
            if (n != 0u) {                                                           // wave-uniform
                const uint32_t dst = s0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                // dst < cap: after a too small hint every slot below the capacity still gets its entry -- the render kernels walk the
                // (clamped) ranges before the call is redone
                if (mine && dst < cap) point_list[dst] = id[rd];
            }
            s0 += n;
        }
    }
}

// Part 1: everything up to the tile ranges and the counters the host reads back (R, C); part 2 (launch_binning_scatter) writes the
// lists.  Split so that the host's read-back can be queued between them and is served while scatter + render still run.
int launch_binning(hipStream_t s, int P, int64_t cap, int gx, int gy, const GeomState& g, const BinState& b, uint32_t* ranges)
{
    const int cgx = (gx + CB - 1) / CB, cgy = (gy + CB - 1) / CB, ncells = cgx * cgy, ntiles = gx * gy;
    const uint32_t* C_dev = g.coffs + P;              // coarse slots in use (null entries included)
    if (cap <= 0) { IBGS_HIP(hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)ntiles, s)); return 0; }      // R = 0 (synchronous sizing)
    const uint32_t ccap = (uint32_t)b.ccap;
    hipLaunchKernelGGL(coarse_emit_kernel, dim3((unsigned)(((size_t)P * 4 + 255) / 256)), dim3(256), 0, s, P, ccap, cgx, (uint32_t)ncells, g.sort_val[0],
                       g.coffs, g.rect, g.tmask, b.ckeys[0], b.cvals[0], b.cid, b.cmask);
    IBGS_HIP(hipGetLastError());
    int bits = 1;
    while ((1 << bits) <= ncells) bits++;             // cell ids 0 .. ncells (ncells = the null key)
    uint32_t* keys[2] = {b.ckeys[0], b.ckeys[1]};
    uint32_t* vals[2] = {b.cvals[0], b.cvals[1]};
    int cur = 0;                                  // which ping-pong buffer holds the sorted entries (no copy back after an odd pass count)
    int rc = radix_sort_pairs(s, keys, vals, (size_t)ccap, bits, b.hist, b.hist_elems, C_dev, false, nullptr, &cur);
    if (rc) return rc;
    const uint32_t* sorted_cells = b.ckeys[cur];
    const uint32_t* sorted_e = b.cvals[cur];
    const unsigned sort_blocks = (unsigned)(((size_t)ccap + 4095) / 4096);          // RS_CHUNK of scan_sort.hip: the histogram's column stride
    hipLaunchKernelGGL(cell_setup_kernel, dim3(1), dim3(256), 0, s, C_dev, ccap, sorted_cells, b.hist, bits <= 8 ? sort_blocks : 0u, ncells,
                       b.cell_start, b.cell_chunk0);
    IBGS_HIP(hipGetLastError());
    const unsigned nchunks_max = (unsigned)(ccap / XCHUNK + (size_t)ncells + 1);
    hipLaunchKernelGGL(expand_count_kernel, dim3(nchunks_max), dim3(64), 0, s, b.cell_start, b.cell_chunk0, ncells, sorted_e, b.cmask, b.chunk_cnt);
    IBGS_HIP(hipGetLastError());
    hipLaunchKernelGGL(cell_scan_kernel, dim3(ncells), dim3(64), 0, s, b.cell_chunk0, ncells, cgx, gx, gy, b.chunk_cnt, b.tile_total);
    IBGS_HIP(hipGetLastError());
    if ((rc = exclusive_scan_u32(s, b.tile_total, b.tile_total, (size_t)ntiles, b.scan_scratch, b.scan_elems, true))) return rc;
    hipLaunchKernelGGL(write_ranges_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, s, ntiles, b.tile_total, ranges, C_dev, g.offsets + P,
                       (uint32_t)(cap < (int64_t)0xFFFFFFFFll ? cap : (int64_t)0xFFFFFFFFll));
    IBGS_HIP(hipGetLastError());
    return cur;                                   // >= 0: the buffer that holds the sorted coarse entries (for part 2)
}

int launch_binning_scatter(hipStream_t s, int64_t cap, int gx, int gy, const BinState& b, int cur)
{
    if (cap <= 0) return 0;
    const int cgx = (gx + CB - 1) / CB, cgy = (gy + CB - 1) / CB, ncells = cgx * cgy;
    const uint32_t* sorted_e = b.cvals[cur];
    const unsigned nchunks_max = (unsigned)(b.ccap / XCHUNK + (size_t)ncells + 1);
    hipLaunchKernelGGL(expand_scatter_kernel, dim3(nchunks_max), dim3(64), 0, s, b.cell_start, b.cell_chunk0, ncells, cgx, gx, gy, sorted_e, b.cid, b.cmask,
                       b.chunk_cnt, b.tile_total, (uint32_t)(cap < (int64_t)0xFFFFFFFFll ? cap : (int64_t)0xFFFFFFFFll), b.point_list);
    IBGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace ibgs
