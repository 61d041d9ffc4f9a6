// A3-A5: per-tile lists (duplicate emission, tile sort, tile ranges) by TWO-LEVEL binning.
//
// The reference emits one 64-bit (tile, depth) key per (Gaussian, tile) pair and sorts all R of them with a 6-pass radix sort
// (DPR/cuda_rasterizer/rasterizer_impl.cu:187-228, 449-457).  Here the P Gaussians are sorted by depth first (scan_sort.hip) and no
// R-sized (nor any other) key array exists after that:
//
//   1. coarse cells of 8 x 8 tiles: a Gaussian has ONE coarse entry per cell that holds at least one of its surviving tiles --
//      (Gaussian id, 64-bit mask of its tiles inside the cell).  C3: 2.6 M entries instead of 12.4 M;
//   2. the entries are PLACED, cell by cell and in depth order inside a cell, by a counting sort over the cells whose digits are
//      recomputed from the Gaussian's rectangle and tile mask (count per block of depth ranks -> scan per cell -> place; below);
//   3. each cell is cut into chunks of 256 entries (one wave each, 4 rounds of 64).  A round is a 64 x 64 bit matrix, row =
//      entry, column = tile of the cell; the wave TRANSPOSES it in registers (wave_bits.h) so that lane t holds which entries
//      go to tile t.  A first kernel counts those bits per chunk; one wave per cell turns the counts into prefixes within the
//      cell and per-tile totals; a scan over the tiles (in global row-major order) gives every tile's range -- `ranges` falls
//      out, no key array, no range-finding pass;
//   4. a second kernel walks each chunk in order and writes the Gaussian ids to their final slots: an entry lane walks the set
//      bits of its mask; slot = start of the tile + prefix of the chunk + entries of earlier rounds + number of earlier lanes
//      in the tile's column.  Entry order inside a tile = order inside the cell = depth order, ties by Gaussian index
//      (SURVEY.md Q9).
//
// The lists are bit-identical to the ones a stable sort of (tile, depth-rank) keys produces (= the reference's lists, restricted
// to the tiles the exact cull keeps); tests compare them with the oracle entry by entry.
#include "common.h"
#include "wave_bits.h"

namespace ibgs {

constexpr int CB = BIN_CELL;            // tiles per cell edge (8)
constexpr int XCHUNK = BIN_XCHUNK;      // coarse entries per expansion chunk = 4 rounds of one wave

// A Gaussian's rectangle and tile mask, loaded once into registers (the mask words beyond the first only exist for rectangles of
// more than 64 tiles, preprocess.hip)
struct RectU { uint32_t x0, x1, y0, y1; };
// (the mask words are four scalars, not an array: selecting a word by a run-time index must stay a chain of register selects -- with an array
// the compiler turns the chain into an indexed load, the whole footprint moves to scratch memory and every row of a large rectangle pays a
// store -> load round trip.  2 % of C3's Gaussians take that path, a third of the waves hold one: cell_count 31 -> 23 us, cell_place 38 -> 28 us)
// masked: the rectangle's tiles are the set bits of the mask (<= IBGS_CULL_MAX_TILES tiles); else rows: the runs of cull_row_run (common.h) are
// recomputed from the Gaussian's record -- rectangles of ANY size are culled exactly --; else every tile of the rectangle (near-singular
// conics, IBGS_FLAG_NO_TILE_CULL: the reference's AABB lists)
struct Footprint { RectU r; uint64_t m0, m1, m2, m3; bool masked, rows; };
static_assert(IBGS_CULL_WORDS == 4, "four mask words");
__device__ __forceinline__ Footprint make_footprint(const uint4 rr, const uint64_t* __restrict__ tmask_hi, uint32_t id)
{
    Footprint f;
    f.r = RectU{rr.x & 0xFFFFu, rr.x >> 16, rr.y & 0xFFFFu, rr.y >> 16};
    const uint32_t area = (f.r.x1 - f.r.x0) * (f.r.y1 - f.r.y0);
    f.masked = area <= (uint32_t)IBGS_CULL_MAX_TILES;
    f.m0 = ((uint64_t)rr.w << 32) | rr.z;
    f.rows = !f.masked && f.m0 == 0ull;                          // (preprocess.hip leaves the mask words of a row-culled large rectangle zero)
    const uint64_t* mw = tmask_hi + (size_t)id * (IBGS_CULL_WORDS - 1);
    f.m1 = (f.masked && area > 64u) ? mw[0] : 0ull;
    f.m2 = (f.masked && area > 128u) ? mw[1] : 0ull;
    f.m3 = (f.masked && area > 192u) ? mw[2] : 0ull;
    return f;
}

// Lane g's footprint, handed to every lane of the wave out of the owner's REGISTERS (round 6).  The cooperative walks below used to load it again from memory --
// wave-uniform addresses, lines the owner had just touched, but a chain of two or three DEPENDENT loads per large Gaussian (footprint -> mask words -> record), ~1 us
// each time, one Gaussian after the other: on a trained scene (a heavy tail of sizes: 1-2 large rectangles in every batch of 64 depth ranks, ten in the unluckiest)
// that chain, not any throughput, was what cell_count / cell_place took 43 / 77 us for instead of 25 / 29 (profiles/r06_frontend.txt).
__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, int g) { return (uint32_t)__builtin_amdgcn_readlane((int)v, g); }
__device__ __forceinline__ uint64_t bcast_u64(uint64_t v, int g) { return ((uint64_t)bcast_u32((uint32_t)(v >> 32), g) << 32) | bcast_u32((uint32_t)v, g); }
__device__ __forceinline__ float bcast_f32(float v, int g) { return __uint_as_float(bcast_u32(__float_as_uint(v), g)); }
__device__ __forceinline__ Footprint bcast_footprint(const Footprint& f, int g)
{
    Footprint o;
    o.r = RectU{bcast_u32(f.r.x0, g), bcast_u32(f.r.x1, g), bcast_u32(f.r.y0, g), bcast_u32(f.r.y1, g)};
    o.m0 = bcast_u64(f.m0, g); o.m1 = bcast_u64(f.m1, g); o.m2 = bcast_u64(f.m2, g); o.m3 = bcast_u64(f.m3, g);
    const uint32_t area = (o.r.x1 - o.r.x0) * (o.r.y1 - o.r.y0);          // (as make_footprint)
    o.masked = area <= (uint32_t)IBGS_CULL_MAX_TILES;
    o.rows = !o.masked && o.m0 == 0ull;
    return o;
}
// ... and what a row-culled rectangle's runs are computed from: quads 0 and 1 of the Gaussian's record (x, y, opacity; conic), loaded by the owner lane
struct RowsRec { float x, y, o, a, b, c; };
__device__ __forceinline__ RowsRec load_rows_rec(const float4* __restrict__ rec, uint32_t id, bool want)
{
    RowsRec r{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (want) { const float4 q0 = rec[(size_t)id * 4], q1 = rec[(size_t)id * 4 + 1]; r = RowsRec{q0.x, q0.y, q0.z, q1.x, q1.y, q1.z}; }
    return r;
}
__device__ __forceinline__ RowsRec bcast_rows_rec(const RowsRec& r, int g) { return RowsRec{bcast_f32(r.x, g), bcast_f32(r.y, g), bcast_f32(r.o, g), bcast_f32(r.a, g), bcast_f32(r.b, g), bcast_f32(r.c, g)}; }

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
//   (cell_setup_block)  first entry of every cell, chunk bookkeeping, C (the number of coarse entries): until round 5 a one-workgroup kernel, now every place
//                       workgroup scans the cells' totals itself and the first one stores the tables
//   cell_place_kernel   same traversal as the count; an entry's slot = first entry of its cell + entries of earlier blocks + entries
//                       of earlier ranks in its own block.  The last term: per batch of 64 ranks (a wave, lane = rank) every cell
//                       collects the lanes that reach it as a 64-bit word in LDS (ds_or); rank inside the batch = set bits below
//                       the own lane, plus the words of the block's earlier waves.
// Cells are handled in slices of at most PLACE_MAX_CELLS (LDS tables); one slice covers a 4K frame.
constexpr int PLACE_THREADS = 256;       // four waves, one batch of 64 consecutive depth ranks each per round
constexpr int PLACE_MAX_CELLS = 1024;

struct PlaceGeom { int P, G, nblk, cgx; int c0, nc; int Pv, gyv; };            // G = depth ranks per block (a multiple of 256); cells [c0, c0 + nc) in this launch;
                                                                               // batched views: instances per view, tile rows per view (one view: P, gy)

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

// ---- large rectangles: the whole wave walks one Gaussian's cells ---------------------------------------------------------------------------
// A lane that walks the 20 .. 135 cells of a large rectangle on its own keeps the other 63 lanes of its batch waiting, and on a trained scene
// (a heavy tail of sizes) nearly every batch of 64 depth ranks holds such a Gaussian: cell_count 23 -> 78 us, cell_place 28 -> 165 us on the
// bench's trained scene.  Rectangles of more than COOP_MIN_CELLS cells (and every row-culled one) are therefore taken out of the per-lane
// path: their owners raise a ballot, and Gaussian after Gaussian the wave loads the footprint again -- wave-uniform addresses, lines the owner
// has just touched -- and gives every lane one cell.  Row-culled rectangles: lane l first computes the run of tile row (strip of 64 rows) + l
// and parks it in LDS; a cell's mask is assembled from its eight rows' runs.
constexpr uint32_t COOP_MIN_CELLS = 9;
__device__ __forceinline__ bool wants_coop(const Footprint& f)
{
    const RectU& r = f.r;
    if (r.x1 <= r.x0 || r.y1 <= r.y0) return false;
    const uint32_t cw = (r.x1 - 1) / CB - r.x0 / CB + 1, chh = (r.y1 - 1) / CB - r.y0 / CB + 1;
    return f.rows || cw * chh > COOP_MIN_CELLS;
}
// the job of a row-culled Gaussian from its record (quads 0, 1: x, y, opacity; conic) -- what preprocess.hip's cull_setup built
__device__ __forceinline__ CullRows rows_job(const PlaceGeom& pg, const Footprint& f, const RowsRec& q, uint32_t id, int& row_off)
{
    CullRows j;
    cull_rows_setup_conic(j, q.x, q.y, q.a, q.b, q.c, cull_qmax(q.o), (int)f.r.x0, (int)f.r.x1);
    row_off = (int)(id / (uint32_t)pg.Pv) * pg.gyv;          // batched views: the record's y is the view's own, the rectangle's rows are the stacked grid's
    return j;
}
// All 64 lanes call this with the same (wave-uniform) footprint; f(cell, mask) runs on the lane that owns the cell.
// CoopCache (round 6): cell_place_kernel walks every large rectangle of a batch TWICE (who touches which cell; then the entries themselves), and on a trained scene
// that walk is what the kernel spends its time on -- ~600 wave instructions per row-culled rectangle and pass (the exact row runs: two square roots and IEEE
// divisions per tile row, then eight rows per cell), 1-2 such rectangles in every batch of 64 depth ranks: 2 370 instructions per wave against 645 on the init
// scene, the kernel VALU-bound chip-wide at 77 us (profiles/r06_frontend.txt).  The first walk therefore parks the mask of every (iteration, lane) in LDS --
// COOP_SLOTS iterations per wave, 2 KB each -- and the second one reads them back instead of computing the runs again; iterations beyond the slots are recomputed.
constexpr int COOP_SLOTS = 8;
struct CoopCache { unsigned long long* base; int slot; int mode; };          // base: [COOP_SLOTS][64] words of this wave; mode 0: none, 1: fill, 2: read
template <typename F>
__device__ __forceinline__ void wave_for_cells(const PlaceGeom& pg, const Footprint& fp, const RowsRec& rq, uint32_t id, int lane,
                                               uint32_t* __restrict__ s_runs /* 64 words of this wave */, F f, CoopCache* cc = nullptr)
{
    const RectU& r = fp.r;
    const uint32_t c0x = r.x0 / CB, c0y = r.y0 / CB, c1x = (r.x1 - 1) / CB, c1y = (r.y1 - 1) / CB;
    const uint32_t cw = c1x - c0x + 1;
    const int mode = cc ? cc->mode : 0;
    // one iteration of the walk: the lane's cell (or none) and its mask -- computed by `compute`, or taken from / left in the cache
    auto iteration = [&](int cell, auto compute) {
        const bool cached = mode != 0 && cc->slot < COOP_SLOTS;          // wave-uniform
        uint64_t m = 0ull;
        if (mode == 2 && cached) m = cc->base[cc->slot * 64 + lane];
        else if (cell >= 0) m = compute();
        if (mode == 1 && cached) cc->base[cc->slot * 64 + lane] = m;
        if (mode != 0) cc->slot++;
        if (cell >= 0 && m != 0ull) f(cell, m);
    };
    if (!fp.rows) {
        const uint32_t ncell = cw * (c1y - c0y + 1);
        for (uint32_t i0 = 0; i0 < ncell; i0 += 64u) {
            const uint32_t idx = i0 + (uint32_t)lane;
            const uint32_t cy = c0y + idx / cw, cx = c0x + idx % cw;
            int cell = idx < ncell ? (int)(cy * (uint32_t)pg.cgx + cx) - pg.c0 : -1;
            if (cell >= pg.nc) cell = -1;
            iteration(cell, [&]() { return cell_mask(fp, cx, cy); });
        }
        return;
    }
    int row_off;
    const CullRows j = rows_job(pg, fp, rq, id, row_off);
    for (uint32_t sy = c0y; sy <= c1y; sy += 8u) {          // strips of eight cell rows = 64 tile rows
        const uint32_t nrow = min(8u, c1y - sy + 1u);
        const int iters = (int)((nrow * cw + 63u) / 64u);
        const bool all_cached = mode == 2 && cc->slot + iters <= COOP_SLOTS;          // wave-uniform: nobody needs this strip's runs
        if (!all_cached) {
            const int ty = (int)(sy * CB) + lane;
            int t0 = 1, t1 = 0;
            if (ty >= (int)r.y0 && ty < (int)r.y1) { if (!cull_row_run(j, ty - row_off, t0, t1)) { t0 = 1; t1 = 0; } }
            __builtin_amdgcn_wave_barrier();
            s_runs[lane] = (uint32_t)t0 | ((uint32_t)t1 << 16);          // (LDS operations of one wave execute in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        for (uint32_t i0 = 0; i0 < nrow * cw; i0 += 64u) {
            const uint32_t idx = i0 + (uint32_t)lane;
            const uint32_t dy = idx / cw, cy = sy + dy, cx = c0x + idx % cw;
            int cell = idx < nrow * cw ? (int)(cy * (uint32_t)pg.cgx + cx) - pg.c0 : -1;
            if (cell >= pg.nc) cell = -1;
            iteration(cell, [&]() {
                const int cx0 = (int)(cx * CB);
                uint64_t m = 0ull;
#pragma unroll
                for (int rr = 0; rr < CB; rr++) {
                    const uint32_t run = s_runs[dy * CB + rr];
                    const int lo = max((int)(run & 0xFFFFu), cx0), hi = min((int)(run >> 16), cx0 + CB - 1);
                    if (lo <= hi) m |= (uint64_t)((1u << (hi - lo + 1)) - 1u) << (rr * CB + (lo - cx0));
                }
                return m;
            });
        }
        if (!all_cached) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- medium rectangles: all of a batch at once, one (Gaussian, cell) item per lane ------------------------------------------------------------------
// A rectangle wider or taller than 8 tiles that stays below the cooperative walk's threshold (<= 9 cells, mask-culled) was walked by its owner lane alone
// (for_cells): up to 9 cells x ~120 instructions while the other lanes of the batch waited.  On the init scene 1 lane in 54 is such a rectangle; on a trained
// scene (log-normal sizes) 3.6 lanes of every batch are, the longest of them 5.5 cells: ~660 wave instructions per walk, three walks per Gaussian (count, touch,
// place) -- 1 320 of the 2 370 instructions a wave of cell_place_kernel executed there (645 on the init scene), the kernel VALU-bound chip-wide at 77 us
// (profiles/r06_frontend.txt).  Here the batch's medium rectangles are flattened: the lanes' cell counts are scanned, item i of the batch = cell k of
// owner o (binary search over the scan with shuffles), every lane fetches ITS owner's footprint out of the owner's registers (ds_bpermute) and computes one
// cell's mask: ~190 wave instructions per 64 items (15 items per batch on the trained scene) instead of 660.
// FlatCache: cell_place_kernel walks the items twice (touch, place); the first walk keeps (cell, mask, owner) of its first FLAT_SLOTS x 64 items in registers.
constexpr int FLAT_SLOTS = 2;
struct FlatCache { int cell[FLAT_SLOTS]; uint64_t m[FLAT_SLOTS]; uint32_t who[FLAT_SLOTS]; /* owner lane | owner id is fetched again: one shuffle */ };
template <int MODE /* 0: no cache, 1: fill, 2: read */, typename F>
__device__ __forceinline__ void wave_flat_cells(const PlaceGeom& pg, const Footprint& fp, uint32_t id, bool medium, int lane, F f /* (cell, mask, owner lane, owner id) */, FlatCache* fc = nullptr)
{
    const RectU& r = fp.r;
    const uint32_t c0x = r.x0 / CB, c0y = r.y0 / CB;
    const uint32_t cw = medium ? (r.x1 - 1) / CB - c0x + 1 : 0u, chh = medium ? (r.y1 - 1) / CB - c0y + 1 : 0u;
    const uint32_t n = cw * chh;
    uint32_t incl = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total == 0u) return;          // wave-uniform
    const uint32_t px = r.x0 | (r.x1 << 16), py = r.y0 | (r.y1 << 16);
    auto item_of = [&](uint32_t base, int& cell, uint64_t& m, uint32_t& owner) {
        const uint32_t item = base + (uint32_t)lane;
        const bool valid = item < total;
        int lo = 0;          // number of lanes whose inclusive count is <= item = the owner (incl is non-decreasing)
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) { const uint32_t v = (uint32_t)__shfl((int)incl, lo + step - 1, 64); if (v <= item) lo += step; }
        owner = (uint32_t)min(lo, 63);
        Footprint fo;
        const uint32_t qx = (uint32_t)__shfl((int)px, (int)owner, 64), qy = (uint32_t)__shfl((int)py, (int)owner, 64);
        fo.r = RectU{qx & 0xFFFFu, qx >> 16, qy & 0xFFFFu, qy >> 16};
        auto sh64 = [&](uint64_t v) { return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)owner, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)v, (int)owner, 64); };
        fo.m0 = sh64(fp.m0); fo.m1 = sh64(fp.m1); fo.m2 = sh64(fp.m2); fo.m3 = sh64(fp.m3);
        const uint32_t excl_o = (uint32_t)__shfl((int)(incl - n), (int)owner, 64), cw_o = (uint32_t)__shfl((int)cw, (int)owner, 64);
        fo.masked = (fo.r.x1 - fo.r.x0) * (fo.r.y1 - fo.r.y0) <= (uint32_t)IBGS_CULL_MAX_TILES;          // (as make_footprint; a medium rectangle is never row-culled)
        fo.rows = false;
        cell = -1; m = 0ull;
        if (valid) {
            const uint32_t k = item - excl_o;
            const uint32_t cx = fo.r.x0 / CB + k % cw_o, cy = fo.r.y0 / CB + k / cw_o;
            const int c = (int)(cy * (uint32_t)pg.cgx + cx) - pg.c0;
            if (c >= 0 && c < pg.nc) { m = cell_mask(fo, cx, cy); if (m != 0ull) cell = c; }
        }
    };
    uint32_t base = 0;
#pragma unroll
    for (int it = 0; it < FLAT_SLOTS; it++) {          // (static indices into the cache: it lives in registers)
        if (base >= total) break;          // wave-uniform
        int cell; uint64_t m; uint32_t owner;
        if (MODE == 2) { cell = fc->cell[it]; m = fc->m[it]; owner = fc->who[it]; }
        else item_of(base, cell, m, owner);
        if (MODE == 1) { fc->cell[it] = cell; fc->m[it] = m; fc->who[it] = owner; }
        const uint32_t oid = (uint32_t)__shfl((int)id, (int)owner, 64);
        if (cell >= 0) f(cell, m, (int)owner, oid);
        base += 64u;
    }
    for (; base < total; base += 64u) {          // more than FLAT_SLOTS x 64 items in one batch: computed in either walk
        int cell; uint64_t m; uint32_t owner;
        item_of(base, cell, m, owner);
        const uint32_t oid = (uint32_t)__shfl((int)id, (int)owner, 64);
        if (cell >= 0) f(cell, m, (int)owner, oid);
    }
}

__global__ void __launch_bounds__(PLACE_THREADS) cell_count_kernel(PlaceGeom pg, const uint32_t* __restrict__ order0, const uint32_t* __restrict__ order1 /* where the depth order lies: n_kept[1] */,
                                                                   const uint32_t* __restrict__ n_kept,
                                                                   const uint4* __restrict__ fpr, const uint64_t* __restrict__ tmask_hi, const float4* __restrict__ rec,
                                                                   uint4* __restrict__ fp_sorted /* the footprints in depth order, for the place kernel */,
                                                                   uint32_t* __restrict__ cnt /* ncells x nblk */)
{
    extern __shared__ uint32_t s_cnt[];
    __shared__ uint32_t s_runs[PLACE_THREADS / 64][64];
    const uint32_t* __restrict__ order = n_kept[1] ? order1 : order0;          // (the depth sort's last pass may have left the result in its input buffer, scan_sort.hip)
    const int tid = threadIdx.x, blk = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < pg.nc; c += PLACE_THREADS) s_cnt[c] = 0u;
    __syncthreads();
    const int j1 = min((int)min((uint32_t)pg.P, *n_kept), (blk + 1) * pg.G);          // ranks past the Gaussians with tiles hold nothing
    for (int j0 = blk * pg.G + wave * 64; j0 < j1; j0 += PLACE_THREADS) {          // wave-uniform: a batch of 64 consecutive ranks per wave
        const int j = j0 + lane;
        const bool have = j < j1;
        bool coop = false, medium = false;
        uint32_t id = 0u;
        Footprint fp{};
        if (have) {
            id = order[j];
            const uint4 rr = fpr[id];                             // the one random 16-byte gather per Gaussian of the binning stage
            if (pg.c0 == 0) fp_sorted[j] = rr;
            fp = make_footprint(rr, tmask_hi, id);
            coop = wants_coop(fp);
            if (!coop) {
                Cells4 c4;
                if (small_cells(pg, fp, c4)) {
#pragma unroll
                    for (int k = 0; k < 4; k++) if (c4.cell[k] >= 0) atomicAdd(&s_cnt[c4.cell[k]], 1u);
                } else medium = true;
            }
        }
        wave_flat_cells<0>(pg, fp, id, medium, lane, [&](int cell, uint64_t, int, uint32_t) { atomicAdd(&s_cnt[cell], 1u); });          // the batch's medium rectangles, an item per lane
        uint64_t todo = __ballot(coop);
        if (todo != 0ull) {
            const RowsRec rq = load_rows_rec(rec, id, coop && fp.rows);          // (every owner at once: one memory round trip for the batch)
            while (todo != 0ull) {          // wave-uniform: the large rectangles of this batch, one after the other, a cell per lane
                const int g = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const uint32_t gid = bcast_u32(id, g);
                const Footprint fg = bcast_footprint(fp, g);
                wave_for_cells(pg, fg, bcast_rows_rec(rq, g), gid, lane, s_runs[wave], [&](int cell, uint64_t) { atomicAdd(&s_cnt[cell], 1u); });
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < pg.nc; c += PLACE_THREADS) cnt[(size_t)(pg.c0 + c) * pg.nblk + blk] = s_cnt[c];
}

// First entry of every cell + chunk bookkeeping + C (ncells <= a few thousand).
// (Round 6 tried to run this as the LAST workgroup of cell_colscan_kernel -- and tile_ranges as the last workgroup of cell_scan_kernel -- behind the classic
// fence / ticket hand-over: two launches fewer, and SLOWER: cell_colscan 7.9 -> 27.4 us, cell_scan 5.0 -> 34.6 us against the 6.5 + 9.5 us of the two
// one-workgroup kernels saved.  The agent-scope release fence every workgroup must issue before its ticket writes the XCD's L2 back; 135 of them in a row
// cost more than a launch.  profiles/r06_frontend.txt.  What did work: no hand-over at all.  Every workgroup of the place kernel scans the cells' totals ITSELF
// (a few hundred words out of L2, cell_starts below) and workgroup 0 also stores the tables the kernels behind it read -- the one-workgroup launch, 6.4 us
// of an idle machine, is gone.)
// The strips: thread t owns cells [t per, (t + 1) per), per = ceil(ncells / 256).  s_part <- exclusive scan of the strips' entry totals; returns the grand total.
__device__ __forceinline__ uint32_t cell_strip_scan(const uint32_t* __restrict__ cell_total, int ncells, uint32_t* s_part /* 256 */, uint32_t* s_w /* 4 */)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (ncells + PLACE_THREADS - 1) / PLACE_THREADS;
    const int c0 = min(ncells, tid * per), c1 = min(ncells, c0 + per);
    uint32_t sum = 0;
    for (int c = c0; c < c1; c++) sum += cell_total[c];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < PLACE_THREADS / 64; w++) { const uint32_t v = s_w[w]; if (w < wave) before += v; all += v; }
    s_part[tid] = before + incl - sum;
    __syncthreads();
    return all;
}
// first entry of cell c, clamped to the capacity: the place kernel drops what does not fit, nobody reads past it, the call is redone (api.hip)
__device__ __forceinline__ uint32_t cell_start_of(const uint32_t* __restrict__ cell_total, int ncells, const uint32_t* s_part, int c, uint32_t ccap)
{
    const int per = (ncells + PLACE_THREADS - 1) / PLACE_THREADS;
    const int t = c / per;
    uint32_t run = s_part[t];
    for (int k = t * per; k < c; k++) run += cell_total[k];
    return min(run, ccap);
}
// ... and the tables for the kernels behind the place kernel (one workgroup): cell_start, the cells' first chunks, C
__device__ __forceinline__ void cell_setup_block(uint32_t ccap, const uint32_t* __restrict__ cell_total, int ncells, const uint32_t* s_part, uint32_t all, uint32_t* s_w,
                                                 uint32_t* __restrict__ cell_start /* ncells + 1 */, uint32_t* __restrict__ cell_chunk0 /* ncells + 1 */, uint32_t* __restrict__ C_out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (ncells + PLACE_THREADS - 1) / PLACE_THREADS;
    const int c0 = min(ncells, tid * per), c1 = min(ncells, c0 + per);
    if (tid == 0) {
        *C_out = all;                                       // entries the frame needs; > ccap: the arena was carved for a too small hint
        cell_start[ncells] = min(all, ccap);
    }
    uint32_t run = s_part[tid], chunks = 0;
    for (int c = c0; c < c1; c++) {
        const uint32_t st = min(run, ccap), en = min(run + cell_total[c], ccap);
        cell_start[c] = st; run += cell_total[c];
        chunks += (en - st + XCHUNK - 1) / XCHUNK;
    }
    uint32_t incl = chunks;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
    __syncthreads();          // (s_w was read by everybody in cell_strip_scan)
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t before = 0, allc = 0;
#pragma unroll
    for (int w = 0; w < PLACE_THREADS / 64; w++) { const uint32_t v = s_w[w]; if (w < wave) before += v; allc += v; }
    if (tid == 0) cell_chunk0[ncells] = allc;
    uint32_t r2 = before + incl - chunks;
    run = s_part[tid];
    for (int c = c0; c < c1; c++) {
        const uint32_t st = min(run, ccap), en = min(run + cell_total[c], ccap);
        cell_chunk0[c] = r2; r2 += (en - st + XCHUNK - 1) / XCHUNK; run += cell_total[c];
    }
}

// One workgroup per cell: exclusive scan of the cell's counts over the blocks, in place, and the cell's total
__global__ void __launch_bounds__(256) cell_colscan_kernel(int nblk, uint32_t* __restrict__ cnt, uint32_t* __restrict__ cell_total)
{
    __shared__ uint32_t s_part[256];
    uint32_t* row = cnt + (size_t)blockIdx.x * nblk;
    const int per = (nblk + 255) / 256;
    const int b0 = min(nblk, (int)threadIdx.x * per), b1 = min(nblk, b0 + per);
    uint32_t sum = 0;
    for (int b = b0; b < b1; b++) sum += row[b];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    // inclusive scan of the 256 strip sums (Hillis-Steele in LDS)
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t v = (threadIdx.x >= (unsigned)d) ? s_part[threadIdx.x - d] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;
    for (int b = b0; b < b1; b++) { const uint32_t v = row[b]; row[b] = run; run += v; }
    if (threadIdx.x == 255) cell_total[blockIdx.x] = s_part[255];
}

// The caller's launch order hint for the blend kernel (ibgs_forward_args::tile_order_hint) is used only when it holds every tile exactly once
// -- whatever else the buffer may contain must not be able to change the image.  Checked against a bitmap of the tiles in LDS, one atomic OR per
// word of the hint, by ONE EXTRA workgroup of the place kernel: beside the placement's thousands of workgroups it costs no time (inside the
// one-workgroup tile_ranges_kernel, where it was first, it added 8 us to the forward).  meta[11] = 1: valid.
constexpr int HINT_MAX_TILES = 65536;
__device__ __forceinline__ void check_order_hint(int ntiles, const uint32_t* __restrict__ order_hint, uint32_t* __restrict__ meta, uint32_t* __restrict__ s_dyn)
{
    // the bitmap lives in the kernel's DYNAMIC LDS (the placement workgroups' tables; the launcher sizes it for whichever is larger): a static
    // 8 KB array would be charged to every workgroup of the kernel for the sake of this one
    uint32_t* s_seen = s_dyn;          // (ntiles + 31) / 32 words
    __shared__ uint32_t s_cnt;
    int ok = 0;
    if (order_hint && ntiles <= HINT_MAX_TILES) {
        const int nslots = (ntiles + ORDER_CLASSES - 1) / ORDER_CLASSES * ORDER_CLASSES;
        for (int w = threadIdx.x; w < (ntiles + 31) / 32; w += PLACE_THREADS) s_seen[w] = 0u;
        if (threadIdx.x == 0) s_cnt = 0u;
        __syncthreads();
        int bad = 0, mine = 0;
        constexpr int BATCH = 16;          // loads in flight per thread before the LDS atomics (one by one the 32 round trips of a 1080p order made this workgroup the kernel's straggler)
        for (int i0 = 0; i0 < nslots; i0 += BATCH * PLACE_THREADS) {
            uint32_t t[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; k++) { const int i = i0 + k * PLACE_THREADS + (int)threadIdx.x; t[k] = i < nslots ? order_hint[i] : 0xFFFFFFFFu; }
#pragma unroll
            for (int k = 0; k < BATCH; k++) {
                if (t[k] == 0xFFFFFFFFu) continue;
                t[k] &= ~ORDER_SPLIT_BIT;          // (the writer's choice of wave shape for the tile: common.h)
                if (t[k] >= (uint32_t)ntiles) { bad = 1; continue; }
                const uint32_t bit = 1u << (t[k] & 31u);
                if (atomicOr(&s_seen[t[k] >> 5], bit) & bit) bad = 1;
                mine++;
            }
        }
        if (mine) atomicAdd(&s_cnt, (uint32_t)mine);
        const int any_bad = __syncthreads_or(bad);          // (also orders the adds to s_cnt before the read)
        ok = (!any_bad && s_cnt == (uint32_t)ntiles) ? 1 : 0;
    }
    if (threadIdx.x == 0) meta[11] = (uint32_t)ok;
}

__global__ void __launch_bounds__(PLACE_THREADS) cell_place_kernel(PlaceGeom pg, uint32_t ccap, const uint32_t* __restrict__ order0, const uint32_t* __restrict__ order1, const uint32_t* __restrict__ n_kept,
                                                                   const uint4* __restrict__ fp_sorted, const uint64_t* __restrict__ tmask_hi, const float4* __restrict__ rec,
                                                                   const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ cell_total, int ncells,
                                                                   uint32_t* __restrict__ cell_start /* written by the first slice's first workgroup */, uint32_t* __restrict__ cell_chunk0, uint32_t* __restrict__ C_out,
                                                                   uint4* __restrict__ cent, int ntiles, const uint32_t* __restrict__ order_hint, uint32_t* __restrict__ meta)
{
    // the extra workgroup (first slice only) is workgroup 0: dispatched first, it runs beside the placement instead of after it
    extern __shared__ unsigned long long s_place[];        // 4 x nc lane words (one table per wave), then nc next-free slots
    if (meta) { if (blockIdx.x == 0) { check_order_hint(ntiles, order_hint, meta, reinterpret_cast<uint32_t*>(s_place)); return; } }
    __shared__ uint32_t s_runs[PLACE_THREADS / 64][64];
    __shared__ unsigned long long s_coop[PLACE_THREADS / 64][COOP_SLOTS * 64];          // the first walk's masks of this wave's large rectangles, for the second one (CoopCache)
    const int nc = pg.nc;
    unsigned long long* s_touch = s_place;
    uint32_t* s_base = reinterpret_cast<uint32_t*>(s_place + 4 * nc);
    const uint32_t* __restrict__ order = n_kept[1] ? order1 : order0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = (int)blockIdx.x - (meta ? 1 : 0);
    __shared__ uint32_t s_part[PLACE_THREADS], s_w4[PLACE_THREADS / 64];
    const uint32_t all = cell_strip_scan(cell_total, ncells, s_part, s_w4);
    for (int c = tid; c < nc; c += PLACE_THREADS) s_base[c] = cell_start_of(cell_total, ncells, s_part, pg.c0 + c, ccap) + cnt[(size_t)(pg.c0 + c) * pg.nblk + blk];
    if (blk == 0 && pg.c0 == 0) cell_setup_block(ccap, cell_total, ncells, s_part, all, s_w4, cell_start, cell_chunk0, C_out);          // (workgroup-uniform)
    const int j1 = min((int)min((uint32_t)pg.P, *n_kept), (blk + 1) * pg.G);
    for (int j0 = blk * pg.G; j0 < j1; j0 += PLACE_THREADS) {          // uniform over the workgroup
        for (int c = tid; c < 4 * nc; c += PLACE_THREADS) s_touch[c] = 0ull;
        __syncthreads();
        const int j = j0 + tid;
        const bool have = j < j1;
        const uint32_t id = have ? order[j] : 0u;
        unsigned long long* mine = s_touch + wave * nc;
        Footprint fp{}; Cells4 c4; bool small = true, coop = false, medium = false;
#pragma unroll
        for (int k = 0; k < 4; k++) c4.cell[k] = -1;
        if (have) {
            fp = make_footprint(fp_sorted[j], tmask_hi, id);
            coop = wants_coop(fp);
            if (!coop) {
                small = small_cells(pg, fp, c4);
                if (small) {
#pragma unroll
                    for (int k = 0; k < 4; k++) if (c4.cell[k] >= 0) atomicOr(&mine[c4.cell[k]], 1ull << lane);
                } else medium = true;
            }
        }
        FlatCache fcache;
        wave_flat_cells<1>(pg, fp, id, medium, lane, [&](int cell, uint64_t, int o, uint32_t) { atomicOr(&mine[cell], 1ull << o); }, &fcache);          // the batch's medium rectangles: who touches which cell ...
        // the large rectangles of this wave's batch, one after the other, a cell per lane (wave_for_cells): first who touches which cell ...
        const uint64_t coopm = __ballot(coop);
        const RowsRec rq = load_rows_rec(rec, id, coop && fp.rows);          // (all owners of the batch at once; out of their registers from here on: bcast_footprint)
        CoopCache cc{s_coop[wave], 0, 1};
        for (uint64_t todo = coopm; todo != 0ull; todo &= todo - 1ull) {
            const int g = __builtin_ctzll(todo);
            const uint32_t gid = bcast_u32(id, g);
            const Footprint fg = bcast_footprint(fp, g);
            wave_for_cells(pg, fg, bcast_rows_rec(rq, g), gid, lane, s_runs[wave], [&](int cell, uint64_t) { atomicOr(&mine[cell], 1ull << g); }, &cc);
        }
        __syncthreads();
        auto place_as = [&](int cell, uint64_t m, uint32_t gid, int g) {          // entry of the batch's rank g
            uint32_t pos = s_base[cell] + (uint32_t)__popcll(mine[cell] & ((1ull << g) - 1ull));
            for (int w = 0; w < wave; w++) pos += (uint32_t)__popcll(s_touch[w * nc + cell]);       // earlier waves = earlier ranks
            if (pos < ccap) cent[pos] = make_uint4(gid, 0u, (uint32_t)m, (uint32_t)(m >> 32));          // one 16-byte store per entry
        };
        auto place = [&](int cell, uint64_t m) { place_as(cell, m, id, lane); };
        if (have && small && !coop) {
#pragma unroll
            for (int k = 0; k < 4; k++) if (c4.cell[k] >= 0) place(c4.cell[k], c4.m[k]);
        }
        wave_flat_cells<2>(pg, fp, id, medium, lane, [&](int cell, uint64_t m, int o, uint32_t oid) { place_as(cell, m, oid, o); }, &fcache);          // ... and their entries
        // ... then the entries themselves
        cc.slot = 0; cc.mode = 2;          // the same walks in the same order: slot k holds what iteration k computed
        for (uint64_t todo = coopm; todo != 0ull; todo &= todo - 1ull) {
            const int g = __builtin_ctzll(todo);
            const uint32_t gid = bcast_u32(id, g);
            const Footprint fg = bcast_footprint(fp, g);
            wave_for_cells(pg, fg, bcast_rows_rec(rq, g), gid, lane, s_runs[wave], [&](int cell, uint64_t m) { place_as(cell, m, gid, g); }, &cc);
        }
        __syncthreads();
        for (int c = tid; c < nc; c += PLACE_THREADS)
            s_base[c] += (uint32_t)(__popcll(s_touch[c]) + __popcll(s_touch[nc + c]) + __popcll(s_touch[2 * nc + c]) + __popcll(s_touch[3 * nc + c]));
        __syncthreads();                                   // (blocks of more than 256 ranks: the next round clears the lane words)
    }
}

// which cell owns chunk `ch` (cell_chunk0 is non-decreasing, cell_chunk0[ncells] = number of chunks)
__device__ __forceinline__ int cell_of_chunk(const uint32_t* __restrict__ cell_chunk0, int ncells, uint32_t ch)
{
    int lo = 0, hi = ncells;          // largest c with cell_chunk0[c] <= ch and cell_chunk0[c + 1] > ch
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cell_chunk0[mid] <= ch) lo = mid; else hi = mid; }
    return lo;
}

// One wave per chunk: counts of entries touching each of the cell's 64 tiles (lane = tile).  A round of 64 entries is a 64 x 64 bit
// matrix (row = entry, column = tile of the cell) with one row per lane; the wave transposes it (wave_bits.h, 31 instructions) so
// that lane t holds column t, and counts its bits -- instead of 64 ballots.
__global__ void __launch_bounds__(64) expand_count_kernel(const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ cell_chunk0, int ncells,
                                                          const uint4* __restrict__ cent,
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
        const uint4 e = (i < i1) ? cent[i] : make_uint4(0u, 0u, 0u, 0u);
        mlo[rd] = e.z; mhi[rd] = e.w;
    }
    const BitTransposeConsts btc = bit_transpose_consts(lane);
    uint32_t cnt = 0;                  // lane t: entries of this chunk whose mask has bit t
#pragma unroll
    for (int rd = 0; rd < XCHUNK / 64; rd++) {
        if (i0 + (uint32_t)rd * 64 >= i1) break;          // wave-uniform
        wave_bit_transpose64(mlo[rd], mhi[rd], btc);
        cnt += (uint32_t)__popc(mlo[rd]) + (uint32_t)__popc(mhi[rd]);
    }
    chunk_cnt[(size_t)ch * 64 + lane] = cnt;
}

// Per-tile totals -> tile starts (exclusive scan, in place; [ntiles] = R) -> ranges, ONE workgroup of 16 waves.
// Empty tiles keep (0, 0) like identifyTileRanges (rasterizer_impl.cu:233-255 after its memset).
constexpr int TR_WAVES = 16;
__global__ void __launch_bounds__(64 * TR_WAVES) tile_ranges_kernel(int ntiles, uint32_t* __restrict__ tile_start /* ntiles + 1: totals in, starts out */,
                                                                   uint32_t* __restrict__ ranges, uint32_t* __restrict__ counters /* R, -, C */, uint32_t cap,
                                                                   const uint32_t* __restrict__ sort_flag /* the depth sort's error word, or nullptr */,
                                                                   uint32_t* __restrict__ host_note /* pinned HOST words or nullptr: [0] R, [2] C, [3] |= sort error, [4] = note_ticket */, uint32_t note_ticket)
{
    constexpr int NW = TR_WAVES;
    __shared__ uint32_t s_wave[NW];
    // Every wave owns a contiguous range of tiles and walks it 64 tiles at a time: coalesced loads and stores, a wave-level scan per step.
    // (A strip of consecutive tiles per THREAD made every lane touch its own cache line, per step: 11 us at 1080p, 54 us for the 32 640 tiles
    // of a four-view batched depth pass.)  The totals stay in registers between the summing pass and the writing pass.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_wave = ((ntiles + NW - 1) / NW + 63) / 64 * 64;          // tiles per wave, a multiple of 64
    const int w0 = min(ntiles, wave * per_wave), w1 = min(ntiles, w0 + per_wave);
    constexpr int KEEP = 32;                               // steps kept in registers: 16 waves x 32 steps x 64 tiles = 32 K tiles (beyond: read twice; 64 steps would spill at 1 024 threads)
    const int nsteps = (w1 - w0 + 63) / 64;
    uint32_t v[KEEP];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < KEEP; k++) { const int t = w0 + k * 64 + lane; v[k] = (k < nsteps && t < w1) ? tile_start[t] : 0u; sum += v[k]; }
    for (int k = KEEP; k < nsteps; k++) { const int t = w0 + k * 64 + lane; sum += t < w1 ? tile_start[t] : 0u; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d, 64);          // the wave's total, in every lane
    if (lane == 0) s_wave[wave] = sum;
    __syncthreads();
    uint32_t run = 0;                                      // tiles in front of this wave's range
    for (int w = 0; w < wave; w++) run += s_wave[w];
    // clamped to the capacity of point_list: after a too small hint the render kernels must not walk past it (the call is redone)
    auto step = [&](int k, uint32_t n) {
        uint32_t inc = n;                                  // inclusive scan over the 64 tiles of the step
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += o; }
        const uint32_t start = run + inc - n;
        const int t = w0 + k * 64 + lane;
        if (t < w1) {
            tile_start[t] = start;
            const uint32_t a = min(start, cap), b = min(start + n, cap);
            *reinterpret_cast<uint2*>(ranges + 2 * (size_t)t) = (b > a) ? make_uint2(a, b) : make_uint2(0u, 0u);
        }
        run += (uint32_t)__shfl((int)inc, 63, 64);
    };
#pragma unroll
    for (int k = 0; k < KEEP; k++) if (k < nsteps) step(k, v[k]);          // (wave-uniform condition)
    for (int k = KEEP; k < nsteps; k++) { const int t = w0 + k * 64 + lane; step(k, t < w1 ? tile_start[t] : 0u); }
    if (tid == NW * 64 - 1) {
        tile_start[ntiles] = run; counters[0] = run;      // R as the binning counted it (the last wave ends at ntiles, whatever its own range)
        // what the host looks at LATER without having queued anything for it (api.hip, round 5: the copies, events and the stream wait that carried these words
        // cost the forward ~10 us of host time): diagnostics and the depth sort's sticky error word, stored straight into pinned host memory
        if (host_note) {
            host_note[0] = run; host_note[2] = counters[2]; if (sort_flag && *sort_flag) host_note[3] = 1u;
            __threadfence_system();          // the words above are visible to the host before the ticket that says "this forward's binning has run" (api.hip: check_sort_flag)
            __hip_atomic_store(host_note + 4, note_ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// One workgroup per cell (lane = tile of the cell, CSCAN_WAVES waves): counts -> exclusive prefixes over the cell's chunks, per-tile totals.
// Every wave takes a contiguous share of the cell's chunks: sums it (eight loads in flight), learns the sums of the waves before it
// through LDS, then walks its share again writing the prefixes (the second read comes out of the L2).  One wave per cell walked a
// crowded cell alone: 26 us instead of 7 when half of the Gaussians sit in one blob (`bench.py --cluster 0.5`).
constexpr int CSCAN_WAVES = 8;
__global__ void __launch_bounds__(64 * CSCAN_WAVES) cell_scan_kernel(const uint32_t* __restrict__ cell_chunk0, int ncells, int cgx, int gx, int gy,
                                                                     uint32_t* __restrict__ chunk_cnt, uint32_t* __restrict__ tile_total /* ntiles */)
{
    __shared__ uint32_t s_sum[CSCAN_WAVES][64];
    const int cell = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c0 = cell_chunk0[cell], c1 = cell_chunk0[cell + 1];
    const uint32_t per = (c1 - c0 + CSCAN_WAVES - 1) / CSCAN_WAVES;
    const uint32_t w0 = min(c1, c0 + (uint32_t)wave * per), w1 = min(c1, w0 + per);
    uint32_t sum = 0;
    for (uint32_t ch = w0; ch < w1; ch += 8) {
        uint32_t c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = (ch + k < w1) ? chunk_cnt[(size_t)(ch + k) * 64 + lane] : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) sum += c[k];
    }
    s_sum[wave][lane] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int w = 0; w < CSCAN_WAVES; w++) { const uint32_t v = s_sum[w][lane]; if (w < wave) run += v; total += v; }
    for (uint32_t ch = w0; ch < w1; ch += 8) {          // eight independent loads in flight, then the serial prefix
        uint32_t c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = (ch + k < w1) ? chunk_cnt[(size_t)(ch + k) * 64 + lane] : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) if (ch + k < w1) { chunk_cnt[(size_t)(ch + k) * 64 + lane] = run; run += c[k]; }
    }
    const int tx = (cell % cgx) * CB + (lane & 7), ty = (cell / cgx) * CB + (lane >> 3);
    if (wave == 0 && tx < gx && ty < gy) tile_total[ty * gx + tx] = total;
}

// One wave per chunk: ids to their final slots, in order.  Per round of 64 entries the wave transposes the bit matrix (row = entry,
// column = tile; wave_bits.h); lane t then holds WHICH entries go to tile t and walks the set bits of that word in ascending order
// (= entry order), round after round, picking each entry's id out of LDS.  The walk takes as many steps as the fullest tile of
// the round has entries (a walk over the ROWS -- every entry lane placing its own id into its tiles -- would take 64 steps whenever
// one large Gaussian covers the cell).  The ids are first gathered in LDS, tile after tile, and then written out with consecutive
// lanes on consecutive slots of one tile's list: a store per (entry, tile) straight from the walk is one partial-line write request
// each and costs 45 us of the kernel's 68 at C3.  A chunk with more than SCAT_STAGE ids (rare: 16 per entry) stores directly.
constexpr int SCAT_STAGE = 4096;
__global__ void __launch_bounds__(64) expand_scatter_kernel(const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ cell_chunk0, int ncells,
                                                            int cgx, int gx, int gy, const uint4* __restrict__ cent,
                                                            const uint32_t* __restrict__ chunk_pref, const uint32_t* __restrict__ tile_start,
                                                            uint32_t cap, uint32_t* __restrict__ point_list)
{
    __shared__ uint32_t s_id[XCHUNK / 64][64];
    __shared__ uint8_t s_out[SCAT_STAGE];              // which entry of the chunk (round << 6 | lane): one byte per id, the id itself stays in s_id
    const uint32_t ch = blockIdx.x;
    if (ch >= cell_chunk0[ncells]) return;
    const int lane = threadIdx.x;
    const int cell = cell_of_chunk(cell_chunk0, ncells, ch);
    const uint32_t i0 = cell_start[cell] + (ch - cell_chunk0[cell]) * XCHUNK, i1 = min(cell_start[cell + 1], i0 + XCHUNK);
    // lane t: first slot of tile t for this chunk
    const int tx = (cell % cgx) * CB + (lane & 7), ty = (cell / cgx) * CB + (lane >> 3);
    const uint32_t slot0 = (tx < gx && ty < gy) ? tile_start[ty * gx + tx] + chunk_pref[(size_t)ch * 64 + lane] : 0u;
    uint32_t mlo[XCHUNK / 64], mhi[XCHUNK / 64];
#pragma unroll
    for (int rd = 0; rd < XCHUNK / 64; rd++) {
        const uint32_t i = i0 + (uint32_t)rd * 64 + lane;
        const uint4 e = (i < i1) ? cent[i] : make_uint4(0u, 0u, 0u, 0u);
        mlo[rd] = e.z; mhi[rd] = e.w;
        s_id[rd][lane] = e.x;
    }
    const BitTransposeConsts btc = bit_transpose_consts(lane);
    uint32_t n = 0;                                        // lane t: ids of this chunk for tile t
#pragma unroll
    for (int rd = 0; rd < XCHUNK / 64; rd++) {
        if (i0 + (uint32_t)rd * 64 < i1) wave_bit_transpose64(mlo[rd], mhi[rd], btc);          // (wave-uniform; rounds past the end hold zeros)
        n += (uint32_t)__popc(mlo[rd]) + (uint32_t)__popc(mhi[rd]);
    }
    uint32_t inc = n;                                      // inclusive scan over the tiles: the chunk's ids, tile after tile
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += o; }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    const bool staged = total <= (uint32_t)SCAT_STAGE;      // wave-uniform
    const uint32_t base = inc - n;
    __syncthreads();                                       // one wave: orders the LDS writes of s_id before the reads below
    uint32_t k = staged ? base : slot0;
#pragma unroll
    for (int rd = 0; rd < XCHUNK / 64; rd++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint32_t bits = half ? mhi[rd] : mlo[rd];
            while (bits != 0u) {
                const int l = __builtin_ctz(bits) + 32 * half;
                bits &= bits - 1u;
                // < cap: after a too small hint every slot below the capacity still gets its entry -- the render kernels walk the
                // (clamped) ranges before the call is redone
                if (staged) s_out[k] = (uint8_t)(rd * 64 + l);
                else if (k < cap) point_list[k] = s_id[rd][l];
                k++;
            }
        }
    }
    if (!staged) return;
    __syncthreads();
    // write-out, four tiles per step: 16 lanes per tile walk its ids
    const int sub = lane & 15, grp = lane >> 4;
    for (int t0 = 0; t0 < 64; t0 += 4) {
        const int t = t0 + grp;
        const uint32_t nt = (uint32_t)__shfl((int)n, t, 64), bt = (uint32_t)__shfl((int)base, t, 64), st = (uint32_t)__shfl((int)slot0, t, 64);
        for (uint32_t q = (uint32_t)sub; q < nt; q += 16u)
            if (st + q < cap) point_list[st + q] = (&s_id[0][0])[s_out[bt + q]];
    }
}

// depth ranks per block of the placement kernels: 256, or more when the count matrix (ncells x blocks) would not fit its share
// of the arena (BinState::carve)
static int place_block_ranks(int P, size_t cnt_elems, int ncells)
{
    size_t max_blocks = cnt_elems / (size_t)(ncells > 0 ? ncells : 1);
    size_t G = 256;
    if (max_blocks == 0) return -1;
    // ... and more as soon as the matrix would pass 2 M counts: it is written by the count pass, scanned column by column and read again by the place
    // pass, 12 bytes of traffic per count.  A batched depth pass over four 1080p views (4 M instances, 540 cells) had 8.4 M of them with blocks
    // of 256: cell_colscan 88 us, cell_count 147 us; with blocks of 1 280 ranks 1.7 M
    const size_t budget = ((size_t)2 << 20) / (size_t)(ncells > 0 ? ncells : 1);
    if (budget >= 1 && budget < max_blocks) max_blocks = budget;
    if (((size_t)P + G - 1) / G > max_blocks) G = (((size_t)P + max_blocks - 1) / max_blocks + 255) / 256 * 256;
    return (int)G;
}

// Part 1: everything up to the tile ranges and the counters the host reads back (R, C); part 2 (launch_binning_scatter) writes the
// lists.  Split so that the host's read-back can be queued between them and is served while scatter + render still run.
int launch_binning(hipStream_t s, int P, int64_t cap, int gx, int gy, const GeomState& g, const BinState& b, uint32_t* ranges,
                   const uint32_t* order_hint, uint32_t* meta, int n_views, const uint32_t* sort_flag, uint32_t* host_note, uint32_t note_ticket)
{
    const int cgx = (gx + CB - 1) / CB, cgy = (gy + CB - 1) / CB, ncells = cgx * cgy, ntiles = gx * gy;
    uint32_t* counters = g.offsets + P;               // R, depth sort error flag, C: what the host reads back in ONE copy (api.hip)
    if (cap <= 0) {      // R = 0 (synchronous sizing): empty ranges, and no hint was looked at
        IBGS_HIP(hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)ntiles, s));
        if (meta) IBGS_HIP(hipMemsetAsync(meta + 11, 0, sizeof(uint32_t), s));
        return 0;
    }
    const uint32_t ccap = (uint32_t)b.ccap;
    PlaceGeom pg;
    pg.P = P; pg.cgx = cgx;
    pg.Pv = P / (n_views > 1 ? n_views : 1); pg.gyv = gy / (n_views > 1 ? n_views : 1);
    const float4* rec = reinterpret_cast<const float4*>(g.rec);
    pg.G = place_block_ranks(P, b.cnt_elems, ncells);
    if (pg.G <= 0) { set_error("binning arena too small for the cell count matrix"); return -IBGS_ERR_ALLOC; }
    pg.nblk = (P + pg.G - 1) / pg.G;
    const uint32_t* order = g.sort_val[0];
    for (pg.c0 = 0; pg.c0 < ncells; pg.c0 += PLACE_MAX_CELLS) {
        pg.nc = min(PLACE_MAX_CELLS, ncells - pg.c0);
        hipLaunchKernelGGL(cell_count_kernel, dim3((unsigned)pg.nblk), dim3(PLACE_THREADS), sizeof(uint32_t) * (size_t)pg.nc, s, pg, order, g.sort_val[1], g.offsets + P + 3, g.fp, g.tmask_hi, rec, g.fp_sorted, b.cnt);
        IBGS_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(cell_colscan_kernel, dim3((unsigned)ncells), dim3(256), 0, s, pg.nblk, b.cnt, b.cell_total);
    IBGS_HIP(hipGetLastError());
    for (pg.c0 = 0; pg.c0 < ncells; pg.c0 += PLACE_MAX_CELLS) {
        pg.nc = min(PLACE_MAX_CELLS, ncells - pg.c0);
        const bool check = meta != nullptr && pg.c0 == 0;
        size_t lds = 36u * (size_t)pg.nc;
        if (check && order_hint && ntiles <= HINT_MAX_TILES) lds = max(lds, sizeof(uint32_t) * (size_t)((ntiles + 31) / 32));
        hipLaunchKernelGGL(cell_place_kernel, dim3((unsigned)pg.nblk + (check ? 1u : 0u)), dim3(PLACE_THREADS), lds, s, pg, ccap, order, g.sort_val[1], g.offsets + P + 3, g.fp_sorted, g.tmask_hi, rec,
                           b.cnt, b.cell_total, ncells, b.cell_start, b.cell_chunk0, counters + 2, b.cent, ntiles, order_hint, check ? meta : nullptr);
        IBGS_HIP(hipGetLastError());
    }
    const unsigned nchunks_max = (unsigned)(ccap / XCHUNK + (size_t)ncells + 1);
    hipLaunchKernelGGL(expand_count_kernel, dim3(nchunks_max), dim3(64), 0, s, b.cell_start, b.cell_chunk0, ncells, b.cent, b.chunk_cnt);
    IBGS_HIP(hipGetLastError());
    hipLaunchKernelGGL(cell_scan_kernel, dim3(ncells), dim3(64 * CSCAN_WAVES), 0, s, b.cell_chunk0, ncells, cgx, gx, gy, b.chunk_cnt, b.tile_total);
    IBGS_HIP(hipGetLastError());
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(1), dim3(64 * TR_WAVES), 0, s, ntiles, b.tile_total, ranges, counters,
                       (uint32_t)(cap < (int64_t)0xFFFFFFFFll ? cap : (int64_t)0xFFFFFFFFll), sort_flag, host_note, note_ticket);
    IBGS_HIP(hipGetLastError());
    return 0;
}

int launch_binning_scatter(hipStream_t s, int64_t cap, int gx, int gy, const BinState& b)
{
    if (cap <= 0) return 0;
    const int cgx = (gx + CB - 1) / CB, cgy = (gy + CB - 1) / CB, ncells = cgx * cgy;
    const unsigned nchunks_max = (unsigned)(b.ccap / XCHUNK + (size_t)ncells + 1);
    hipLaunchKernelGGL(expand_scatter_kernel, dim3(nchunks_max), dim3(64), 0, s, b.cell_start, b.cell_chunk0, ncells, cgx, gx, gy, b.cent,
                       b.chunk_cnt, b.tile_total, (uint32_t)(cap < (int64_t)0xFFFFFFFFll ? cap : (int64_t)0xFFFFFFFFll), b.point_list);
    IBGS_HIP(hipGetLastError());
    return 0;
}

}  // namespace ibgs
