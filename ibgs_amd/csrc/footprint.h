// The footprint of a Gaussian on the tile grid (rectangle + mask of surviving tiles, preprocess.hip) and the coarse cells of 8 x 8 tiles
// it reaches: shared by the binning kernels (binning.hip) and by the depth sort's histogram kernel, which counts the entries per cell on
// its way over the keys (scan_sort.hip).
#pragma once
#include "common.h"

namespace ibgs {

constexpr int CB = BIN_CELL;            // tiles per cell edge (8)
constexpr int XCHUNK = BIN_XCHUNK;      // coarse entries per expansion chunk = 4 rounds of one wave

// A Gaussian's rectangle and tile mask, loaded once into registers (the mask words beyond the first only exist for rectangles of
// more than 64 tiles, preprocess.hip)
struct RectU { uint32_t x0, x1, y0, y1; };
// (the mask words are four scalars, not an array: selecting a word by a run-time index must stay a chain of register selects -- with an array
// the compiler turns the chain into an indexed load, the whole footprint moves to scratch memory and every row of a large rectangle pays a
// store -> load round trip: 11 us per round of 512 ranks at C3 before this, round 3)
struct Footprint { RectU r; uint64_t m0, m1, m2, m3; bool masked; };
static_assert(IBGS_CULL_WORDS == 4, "four mask words");
__device__ __forceinline__ Footprint make_footprint(const uint4 rr, const uint64_t* __restrict__ tmask_hi, uint32_t id)
{
    Footprint f;
    f.r = RectU{rr.x & 0xFFFFu, rr.x >> 16, rr.y & 0xFFFFu, rr.y >> 16};
    const uint32_t area = (f.r.x1 - f.r.x0) * (f.r.y1 - f.r.y0);
    f.masked = area <= (uint32_t)IBGS_CULL_MAX_TILES;            // larger rectangles keep every tile (preprocess.hip)
    f.m0 = ((uint64_t)rr.w << 32) | rr.z;
    const uint64_t* mw = tmask_hi + (size_t)id * (IBGS_CULL_WORDS - 1);
    f.m1 = (f.masked && area > 64u) ? mw[0] : 0ull;
    f.m2 = (f.masked && area > 128u) ? mw[1] : 0ull;
    f.m3 = (f.masked && area > 192u) ? mw[2] : 0ull;
    return f;
}

// up to 8 bits of the row-major tile mask, starting at bit `start`
__device__ __forceinline__ uint32_t mask_bits(const Footprint& f, uint32_t start, uint32_t len)
{
    const uint32_t w = start >> 6, o = start & 63u;
    const uint64_t a0 = f.m0, a1 = f.m1, a2 = f.m2, a3 = f.m3;          // values, not addresses
    const uint64_t lo = w == 0 ? a0 : (w == 1 ? a1 : (w == 2 ? a2 : a3));
    const uint64_t hi = w == 0 ? a1 : (w == 1 ? a2 : a3);          // (only used when the run crosses into the next word)
    uint64_t v = lo >> o;
    if (o + len > 64u) v |= hi << (64u - o);
    return (uint32_t)v & ((1u << len) - 1u);
}

// surviving tiles of one Gaussian inside cell (ccx, ccy): bit ly * 8 + lx for tile (8 ccx + lx, 8 ccy + ly)
__device__ __forceinline__ uint64_t cell_mask(const Footprint& f, uint32_t ccx, uint32_t ccy)
{
    const RectU& r = f.r;
    const uint32_t cx0 = ccx * CB, cy0 = ccy * CB;
    const uint32_t xa = max(r.x0, cx0), xb = min(r.x1, cx0 + CB), ya = max(r.y0, cy0), yb = min(r.y1, cy0 + CB);
    if (xa >= xb || ya >= yb) return 0ull;
    const uint32_t w = r.x1 - r.x0;
    const uint32_t len = xb - xa;
    uint64_t m = 0ull;
    for (uint32_t ty = ya; ty < yb; ty++) {
        const uint32_t bits = f.masked ? mask_bits(f, (ty - r.y0) * w + (xa - r.x0), len) : ((1u << len) - 1u);
        m |= (uint64_t)bits << ((ty - cy0) * CB + (xa - cx0));
    }
    return m;
}

// ---- cells in depth order by DIRECT PLACEMENT ------------------------------------------------------------------------------------
// What a stable sort of (cell, depth rank) keys would produce, without materialising keys: a counting sort whose digits (the cells
// a Gaussian reaches) are recomputed from the Gaussian's rectangle and tile mask instead of being read from an array.
//   cell_count_kernel   one workgroup per block of G consecutive depth ranks: entries per cell (LDS histogram) -> cnt[cell][block]
//   cell_colscan_kernel one workgroup per cell: exclusive scan over the blocks, in place; the cell's total
//   cell_setup_kernel   one workgroup: first entry of every cell, chunk bookkeeping, C (the number of coarse entries)
//   cell_place_kernel   same traversal as the count; an entry's slot = first entry of its cell + entries of earlier blocks + entries
//                       of earlier ranks in its own block.  The last term: per batch of 64 ranks (a wave, lane = rank) every cell
//                       collects the lanes that reach it as a 64-bit word in LDS (ds_or); rank inside the batch = set bits below
//                       the own lane, plus the words of the block's earlier waves.
// Cells are handled in slices of at most PLACE_MAX_CELLS (LDS tables); one slice covers a 4K frame.
constexpr int PLACE_THREADS = 256;       // four waves, one batch of 64 consecutive depth ranks each per round

struct PlaceGeom { int P, G, nblk, cgx; int c0, nc; };            // G = depth ranks per block (a multiple of 256); cells [c0, c0 + nc) in this launch

// The common case: a rectangle of at most 8 x 8 tiles reaches at most 2 x 2 cells and its whole mask is word 0.  The rows are spread
// to a stride of 8 once; the part inside a cell is that image shifted by the rectangle's offset from the cell, columns that wrap
// masked off.  Four fixed slots (cell < 0: none), so the callers run straight-line code and keep the masks between their sweeps.
struct Cells4 { int cell[4]; uint64_t m[4]; };
__device__ __forceinline__ bool small_cells(const PlaceGeom& pg, const Footprint& fp, Cells4& out)
{
    const RectU& r = fp.r;
    const uint32_t w = r.x1 - r.x0, h = r.y1 - r.y0;          // (unsigned: an empty, culled rectangle fails the test below or yields no cell)
#pragma unroll
    for (int k = 0; k < 4; k++) { out.cell[k] = -1; out.m[k] = 0ull; }
    if (r.x1 <= r.x0 || r.y1 <= r.y0) return true;            // culled: no tiles
    if (w > (uint32_t)CB || h > (uint32_t)CB) return false;
    const uint32_t c0x = r.x0 / CB, c0y = r.y0 / CB, c1x = (r.x1 - 1) / CB, c1y = (r.y1 - 1) / CB;
    uint64_t img = 0ull;
    const uint64_t rowm = (1ull << w) - 1ull;
    for (uint32_t i = 0; i < h; i++) img |= ((fp.m0 >> (i * w)) & rowm) << (8u * i);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t cx = c0x + (uint32_t)(k & 1), cy = c0y + (uint32_t)(k >> 1);
        if (cx > c1x || cy > c1y) continue;
        const int cell = (int)(cy * (uint32_t)pg.cgx + cx) - pg.c0;
        if (cell < 0 || cell >= pg.nc) continue;
        const int dx = (int)r.x0 - (int)(cx * CB), dy = (int)r.y0 - (int)(cy * CB);          // both in (-8, 8)
        uint64_t m = dx >= 0 ? (img << dx) & (0x0101010101010101ull * (uint64_t)((0xFFu << dx) & 0xFFu))
                             : (img >> (-dx)) & (0x0101010101010101ull * (uint64_t)(0xFFu >> (-dx)));
        m = dy >= 0 ? m << (8 * dy) : m >> (8 * (-dy));
        if (m != 0ull) { out.cell[k] = cell; out.m[k] = m; }
    }
    return true;
}

// larger rectangles: calls f(cell, mask) for every cell of the slice that holds a surviving tile of the Gaussian
template <typename F>
__device__ __forceinline__ void for_cells(const PlaceGeom& pg, const Footprint& fp, F f)
{
    const RectU& r = fp.r;
    const uint32_t c0x = r.x0 / CB, c0y = r.y0 / CB, c1x = (r.x1 - 1) / CB, c1y = (r.y1 - 1) / CB;
    for (uint32_t cy = c0y; cy <= c1y; cy++)
        for (uint32_t cx = c0x; cx <= c1x; cx++) {
            const int cell = (int)(cy * (uint32_t)pg.cgx + cx) - pg.c0;
            if (cell < 0 || cell >= pg.nc) continue;
            const uint64_t m = cell_mask(fp, cx, cy);
            if (m != 0ull) f(cell, m);
        }
}

}  // namespace ibgs
