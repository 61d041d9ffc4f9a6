// C-ABI entry points and stage orchestration (include/ibgs_rast.h).
//
// Stage order of one forward (compare CudaRasterizer::Rasterizer::forward,
// DPR/cuda_rasterizer/rasterizer_impl.cu:320-515):
//   preprocess -> depth sort of the P Gaussians (4 x 8-bit LSD passes) -> tiles-touched scan in
//   depth order -> R read-back (the only host sync, as in the reference :430) -> load-balanced
//   duplicate emission -> stable tile-id sort (ceil(bits/8) passes) -> tile ranges -> render.
// Everything is enqueued on the caller's stream; no device-wide synchronisation, no allocation
// (arenas are caller-owned), no persistent library state.
#include "common.h"
#include <sched.h>
#include <time.h>
#include <string.h>
#include <cstdlib>
#include <stdlib.h>
#include <stdio.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include <map>
#include <mutex>

namespace ibgs {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static uint32_t higher_msb(uint32_t n)
{   // rasterizer_impl.cu:152-167
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

GeomState GeomState::carve(char* base, size_t P, size_t* total)
{
    Carver c(base);
    GeomState g;
    g.rec = c.take<float>(P * REC_FLOATS);
    g.depths = c.take<float>(P);
    g.cov3D = c.take<float>(P * 6);
    g.tiles = c.take<uint32_t>(P);
    g.fp = c.take<uint4>(P);
    g.tmask_hi = c.take<uint64_t>(P * (IBGS_CULL_WORDS - 1));
    g.fp_sorted = c.take<uint4>(P);
    g.clamped = c.take<uint8_t>(P);
    g.alive64 = c.take<uint64_t>((P + 63) / 64);
    g.tile_partial = c.take<uint32_t>((P + 63) / 64 + IBGS_MAX_VIEWS + 1);
    g.sort_key[0] = c.take<uint32_t>(P); g.sort_key[1] = c.take<uint32_t>(P);
    g.sort_val[0] = c.take<uint32_t>(P); g.sort_val[1] = c.take<uint32_t>(P);
    g.offsets = c.take<uint32_t>(P + 5);
    g.hist_elems = radix_hist_elems(P);
    g.hist = c.take<uint32_t>(g.hist_elems);
    if (total) *total = (size_t)(c.cur - reinterpret_cast<uintptr_t>(base)) + 128;
    return g;
}

ImgState ImgState::carve(char* base, int W, int H, size_t* total)
{
    Carver c(base);
    ImgState im;
    const size_t HW = (size_t)W * H;
    const size_t tiles = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    im.ranges = c.take<uint32_t>(tiles * 2);
    im.final_T = c.take<float>(HW);
    im.n_contrib = c.take<uint32_t>(HW);
    im.sum_w = c.take<float>(HW);
    im.low_high = c.take<uint32_t>(HW * 2);
    im.valid_idx = c.take<int32_t>(HW * IBGS_MAX_SRC);
    im.valid_w = c.take<float>(HW * IBGS_MAX_SRC);
    im.meta = c.take<uint32_t>(32);
    im.slot_c = c.take<uint32_t>(HW * IBGS_MAX_BUFFER_LENGTH);
    im.tile_walked = c.take<uint32_t>(tiles * 4);
    im.tile_order = c.take<uint32_t>((tiles + ORDER_CLASSES - 1) / ORDER_CLASSES * ORDER_CLASSES);
    im.tile_risky = c.take<uint32_t>(tiles * 4);
    if (total) *total = (size_t)(c.cur - reinterpret_cast<uintptr_t>(base)) + 128;
    return im;
}

BinState BinState::carve(char* base, size_t R, int W, int H, size_t* total)
{
    Carver c(base);
    BinState b;
    const size_t gx = (size_t)((W + TILE - 1) / TILE), gy = (size_t)((H + TILE - 1) / TILE), ntiles = gx * gy;
    const size_t ncells = ((gx + BIN_CELL - 1) / BIN_CELL) * ((gy + BIN_CELL - 1) / BIN_CELL);
    // The sorted Gaussian ids come first: their offset does not depend on R, so ibgs_backward finds them
    // whether the forward carved the arena for the exact R or for a larger rendered_hint.
    b.point_list = c.take<uint32_t>(R);
    b.ccap = R;                                     // a coarse entry stands for at least one tile entry: C <= R
    b.cent = c.take<uint4>(R);
    // count matrix of the placement (binning.hip): one column per block of >= 256 depth ranks.  Its share of the arena follows R,
    // the only size this function knows; the launcher makes the blocks larger when P / 256 columns do not fit
    b.cnt_elems = (R / 4 > (size_t)65536 ? R / 4 : (size_t)65536) + ncells;
    b.cnt = c.take<uint32_t>(b.cnt_elems);
    b.cell_total = c.take<uint32_t>(ncells);
    b.cell_start = c.take<uint32_t>(ncells + 1);
    b.cell_chunk0 = c.take<uint32_t>(ncells + 1);
    b.chunk_cnt = c.take<uint32_t>((R / BIN_XCHUNK + ncells + 1) * 64);
    b.tile_total = c.take<uint32_t>(ntiles + 1);
    b.scan_elems = scan_scratch_elems(ntiles + 1);
    b.scan_scratch = c.take<uint32_t>(b.scan_elems);
    if (total) *total = (size_t)(c.cur - reinterpret_cast<uintptr_t>(base)) + 128;
    return b;
}

// ---- optional stage timing (bench.py roofline) ------------------------------------------------
struct TimedPair { hipEvent_t a, b; int stage; };
static uint32_t g_time_mask = 0;
static std::vector<TimedPair> g_pairs;
static std::vector<hipEvent_t> g_free_events;

static hipEvent_t get_event()
{
    if (!g_free_events.empty()) { hipEvent_t e = g_free_events.back(); g_free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

StageTimer::StageTimer(hipStream_t s_, int stage_) : s(s_), stage(stage_), on((g_time_mask >> stage_) & 1u)
{
    if (on) { a = get_event(); b = get_event(); (void)hipEventRecord(a, s); }
}
StageTimer::~StageTimer()
{
    if (on) { (void)hipEventRecord(b, s); g_pairs.push_back({a, b, stage}); }
}

static int stage_check(hipStream_t s, bool debug, const char* what)
{
    if (!debug) return 0;
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { set_error("stage '%s' failed: %s", what, hipGetErrorString(e)); return -IBGS_ERR_HIP; }
    return 0;
}

// R of a hinted forward without a copy, an event or a second stream (round 5): ONE workgroup adds up the per-wave tile sums the preprocess kernel left and stores
// the total, then a ticket, straight into pinned host memory; the host polls the ticket (ibgs_forward).  The four runtime calls it replaces -- event
// record, stream wait, copy, event record -- cost the forward ~12 us of host time, and the copy engine's own latency on top.  The ticket is a kernel argument
// (round 6: until round 5 a device word counted it, for a replay from a hipGraph that no longer exists; an argument cannot get out of step after a failed call).
__global__ void __launch_bounds__(1024) rendered_note_kernel(RenderedNote n)
{
    __shared__ unsigned long long s_w[16];
    rendered_note_block<1024>(n, s_w);
}

}  // namespace ibgs

using namespace ibgs;

extern "C" {

const char* ibgs_last_error(void) { return g_err; }
const char* ibgs_version(void) { return "ibgs_rast 0.1 (gfx950)"; }
void ibgs_timing_enable(uint32_t stage_mask) { g_time_mask = stage_mask; }
int32_t ibgs_timing_collect(float* ms, int32_t* launches)
{
    for (int i = 0; i < IBGS_NUM_STAGES; i++) { if (ms) ms[i] = 0.f; if (launches) launches[i] = 0; }
    for (auto& p : g_pairs) {
        float t = 0.f;
        hipError_t e = hipEventSynchronize(p.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, p.a, p.b);
        if (e != hipSuccess) { set_error("timing collect: %s", hipGetErrorString(e)); return -IBGS_ERR_HIP; }
        if (ms) ms[p.stage] += t;
        if (launches) launches[p.stage] += 1;
        g_free_events.push_back(p.a); g_free_events.push_back(p.b);
    }
    g_pairs.clear();
    return 0;
}
size_t ibgs_sizeof_forward_args(void) { return sizeof(ibgs_forward_args); }
size_t ibgs_sizeof_backward_args(void) { return sizeof(ibgs_backward_args); }

size_t ibgs_required_geom(int32_t P) { size_t t; GeomState::carve(nullptr, (size_t)(P > 0 ? P : 0), &t); return t; }
size_t ibgs_required_img(int32_t W, int32_t H) { size_t t; ImgState::carve(nullptr, W, H, &t); return t; }
size_t ibgs_required_binning(int64_t R, int32_t W, int32_t H) { size_t t; BinState::carve(nullptr, (size_t)(R > 0 ? R : 0), W, H, &t); return t; }
size_t ibgs_required_deterministic(int64_t R, int32_t P) { size_t t; DetState::carve(nullptr, (size_t)(R > 0 ? R : 0) * 4, (size_t)(P > 0 ? P : 0), &t); return t; }   // x 4: up to four waves per tile
size_t ibgs_required_tex(int32_t n_src, int32_t W, int32_t H) { return (size_t)n_src * W * H * sizeof(float4) + 128; }
size_t ibgs_required_geo_table(int32_t W, int32_t H) { return geo_table_floats(W, H) * sizeof(float) + 128; }
size_t ibgs_required_geo_table_for(int32_t W, int32_t H, int32_t L)
{   // the window pass writes at most L entries and a terminator in slot L (render_bwd.hip); the blend loop reads no further
    const int slots = (L >= 1 && L < IBGS_MAX_BUFFER_LENGTH) ? L + 1 : IBGS_MAX_BUFFER_LENGTH;
    return (size_t)W * H * slots * GEO_TAB_FIELDS * sizeof(float) + 128;
}
size_t ibgs_required_deterministic_for(int64_t R, int32_t P, int32_t W, int32_t H, int32_t render_geo, uint32_t flags)
{
    ibgs_backward_args a{}; a.W = W; a.H = H; a.render_geo = render_geo; a.flags = flags;
    size_t t; DetState::carve(nullptr, (size_t)(R > 0 ? R : 0) * render_backward_waves_per_tile(a), (size_t)(P > 0 ? P : 0), &t); return t;
}

#define OFF(base_struct, field) if (!strcmp(name, #field)) return (int64_t)((char*)base_struct.field - (char*)nullptr)
int64_t ibgs_geom_offset(int32_t P, const char* name)
{
    size_t t; GeomState g = GeomState::carve(nullptr, (size_t)P, &t);
    OFF(g, rec); OFF(g, depths); OFF(g, cov3D); OFF(g, tiles); OFF(g, fp); OFF(g, tmask_hi); OFF(g, clamped); OFF(g, offsets);
    // the depth order lies in buffer 0 unless the word offsets[P + 4] is 1 (the sort's last pass found every key in one bucket and left its input
    // where it was: scan_sort.hip) -- then "order_alt" / "sorted_depth_keys_alt" hold it; offsets[P + 3] = how many leading entries are valid
    if (!strcmp(name, "order")) return (int64_t)((char*)g.sort_val[0] - (char*)nullptr);
    if (!strcmp(name, "order_alt")) return (int64_t)((char*)g.sort_val[1] - (char*)nullptr);
    if (!strcmp(name, "sorted_depth_keys")) return (int64_t)((char*)g.sort_key[0] - (char*)nullptr);
    if (!strcmp(name, "sorted_depth_keys_alt")) return (int64_t)((char*)g.sort_key[1] - (char*)nullptr);
    return -1;
}
size_t ibgs_tile_order_slots(int32_t W, int32_t H)
{
    const size_t tiles = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    return (tiles + ORDER_CLASSES - 1) / ORDER_CLASSES * ORDER_CLASSES;
}
int64_t ibgs_img_offset(int32_t W, int32_t H, const char* name)
{
    size_t t; ImgState im = ImgState::carve(nullptr, W, H, &t);
    OFF(im, ranges); OFF(im, final_T); OFF(im, n_contrib); OFF(im, sum_w); OFF(im, low_high); OFF(im, valid_idx); OFF(im, valid_w); OFF(im, slot_c); OFF(im, meta); OFF(im, tile_walked); OFF(im, tile_order); OFF(im, tile_risky);
    return -1;
}
int64_t ibgs_binning_offset(int64_t R, int32_t W, int32_t H, const char* name)
{
    size_t t; BinState b = BinState::carve(nullptr, (size_t)R, W, H, &t);
    OFF(b, point_list);
    return -1;
}
#undef OFF

// Pinned host words + event for the R read-back, one per (device, stream) of the PROCESS (created on first use, never freed).  Keyed by the stream, not by the
// calling thread (round 6): PyTorch runs the backward on an autograd worker thread, and the backward must find the slot of the forward it belongs to -- which it
// does through the stream both were queued on.  (Two threads driving one stream at the same time would share a slot: they would be racing on the stream anyway.)
struct RSlot {
    uint32_t* host; hipEvent_t ev;             // 64 bytes of pinned, host-coherent memory: [0..1] R, [2] the ticket of rendered_note_kernel; stats = host + 8
    uint32_t* stats; bool stats_pending;       // stats[0] R as the binning counted it, [2] C, [3] the depth sort's STICKY error word, [4] the ticket of the binning's note: all stored by
                                               // tile_ranges_kernel of a hinted forward, read when somebody asks (stats) or at the next point somebody is waiting anyway (the error word)
    bool flag_pending;                         // a hinted forward whose depth sort nobody has vouched for yet
    hipStream_t last_stream;
    uint32_t seq;                              // tickets handed out on this slot; the last hinted forward's = what stats[4] shows once its binning has run
};
static std::mutex g_slot_mutex;
static std::map<std::pair<int, hipStream_t>, RSlot*> g_slots;
static RSlot* rslot(hipStream_t s, bool create)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("hipGetDevice failed"); return nullptr; }
    std::lock_guard<std::mutex> lock(g_slot_mutex);
    auto it = g_slots.find({dev, s});
    if (it != g_slots.end()) return it->second;
    if (!create) return nullptr;
    RSlot* slot = new RSlot();
    void* p = nullptr;
    // host-coherent and mapped, whatever HIP_HOST_COHERENT says: the device stores into it from running kernels and the host polls it
    if (hipHostMalloc(&p, 64, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { delete slot; set_error("hipHostMalloc for the R read-back failed"); return nullptr; }
    if (hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(p); delete slot; set_error("hipEventCreate failed"); return nullptr; }
    slot->host = static_cast<uint32_t*>(p);
    memset(p, 0, 64);          // (stats[3] is a sticky error word the GPU only ever sets; host[2] / stats[4] the last tickets)
    slot->stats = slot->host + 8;
    slot->stats_pending = false; slot->flag_pending = false; slot->last_stream = s; slot->seq = 0;
    g_slots[{dev, s}] = slot;
    return slot;
}
// The depth sort's error word of a hinted forward is stored into pinned host memory by the binning's last kernel (tile_ranges_kernel), together with the
// diagnostics and -- behind a system-scope fence -- that forward's ticket; nobody queues anything for it and nobody waits for it inside ibgs_forward (round 5: the
// tiles-touched sums leave right after the preprocess kernel, so that the host is released while the GPU is still sorting).  The word is sticky.  It is reported
//   * by the ibgs_backward of the SAME step (round 6): at its end, with every kernel of the backward queued, it waits -- briefly, bounded -- for the binning's note
//     of its forward (normally long there: the loss and the backward's own launches have passed) and fails the call if the word is set: the optimiser never sees
//     gradients of mis-ordered lists;
//   * by the next ibgs_forward on the stream after its own wait for R (a forward that no backward followed), and by ibgs_check_async (eval-only callers).
// (The look-back of the onesweep passes times out only if a workgroup is starved for seconds.)
static int check_sort_flag(RSlot* rs, bool wait_for_note)
{
    if (!rs || !rs->flag_pending) return 0;
    volatile uint32_t* note = rs->stats + 4;
    bool final_ = __atomic_load_n(const_cast<uint32_t*>(note), __ATOMIC_ACQUIRE) == rs->seq;
    if (!final_ && wait_for_note) {
        // bounded: ~2 ms of polling, then the check stays with the next entry into the library (an asynchronous error, like HIP's own)
        struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
        for (uint32_t spins = 1; !final_; spins++) {
            final_ = __atomic_load_n(const_cast<uint32_t*>(note), __ATOMIC_ACQUIRE) == rs->seq;
            if ((spins & 255u) == 0) {
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 2000000ll) break;
            }
        }
    }
    if (*(volatile uint32_t*)(rs->stats + 3)) {
        rs->stats[3] = 0u; rs->flag_pending = false;
        set_error("depth sort: decoupled look-back timed out (the lists of this stream's last forward were mis-ordered)"); return -IBGS_ERR_HIP;
    }
    if (final_) rs->flag_pending = false;          // that forward's binning has run and left the word clear: vouched for
    return 0;
}

// what the last ibgs_forward of this thread saw (bench / tests): R, coarse binning entries C (deferred sizing only, else -1), whether the
// rendered_hint was too small and binning + render ran twice
static thread_local int64_t g_last_stats[3] = {0, -1, 0};
static thread_local RSlot* g_stats_slot = nullptr;
void ibgs_last_forward_stats(int64_t* out)
{
    if (g_stats_slot && g_stats_slot->stats_pending) {          // the coarse count sits in pinned host memory once the binning of that forward has run
        if (hipStreamSynchronize(g_stats_slot->last_stream) == hipSuccess) g_last_stats[1] = (int64_t)*(volatile uint32_t*)(g_stats_slot->stats + 2);
        g_stats_slot->stats_pending = false;
    }
    out[0] = g_last_stats[0]; out[1] = g_last_stats[1]; out[2] = g_last_stats[2];
}
// The asynchronous error of the forwards queued on `stream` (the depth sort's look-back time-out, see check_sort_flag), for callers no backward follows
// (evaluation renders, the last forward of a run).  wait != 0: the stream is drained first, the answer is then final; else only what has arrived is reported.
// tests only: the look-back patience of the depth sort's single-launch passes (0 = give up at the first unpublished word: provokes the asynchronous error path)
void ibgs_debug_set_lookback_spins(uint32_t v) { radix_set_lookback_spins(v); }
int32_t ibgs_check_async(void* stream, int32_t wait)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    RSlot* rs = rslot(s, false);
    if (!rs || !rs->flag_pending) return 0;
    if (wait) IBGS_HIP(hipStreamSynchronize(s));
    return check_sort_flag(rs, false);
}

int64_t ibgs_forward(const ibgs_forward_args* ap)
{
    if (!ap) { set_error("null args"); return -IBGS_ERR_INVALID; }
    const ibgs_forward_args& a = *ap;
    hipStream_t s = reinterpret_cast<hipStream_t>(a.stream);
    const bool debug = (a.flags & IBGS_FLAG_DEBUG) != 0;
    if (a.P < 0 || a.W <= 0 || a.H <= 0) { set_error("bad sizes P=%d W=%d H=%d", a.P, a.W, a.H); return -IBGS_ERR_INVALID; }
    if (a.P == 0) return 0;                                     // rasterize_points.cu:101-102
    if (!a.means3D || !a.opacities || !a.viewmatrix || !a.projmatrix || !a.campos || !a.bg || !a.radii) {
        set_error("missing required pointer"); return -IBGS_ERR_INVALID;
    }
    if ((a.shs == nullptr) == (a.colors_precomp == nullptr) && !a.render_depth_only) {
        set_error("provide exactly one of shs / colors_precomp"); return -IBGS_ERR_INVALID;
    }
    if (((a.scales == nullptr) || (a.rotations == nullptr)) == (a.cov3D_precomp == nullptr)) {
        set_error("provide exactly one of scales+rotations / cov3D_precomp"); return -IBGS_ERR_INVALID;
    }
    if (a.shs && (a.D < 0 || a.D > 3 || (a.D + 1) * (a.D + 1) > a.M)) { set_error("bad SH degree D=%d M=%d", a.D, a.M); return -IBGS_ERR_INVALID; }
    // rows of 16 coefficients are fetched with 16-byte loads (preprocess.hip, preprocess_bwd.hip): the base must be 16-byte aligned (any torch
    // allocation is; a view that starts in the middle of one may not be)
    if (a.shs && a.M == 16 && (reinterpret_cast<uintptr_t>(a.shs) & 15u)) { set_error("shs (M = 16) must be 16-byte aligned"); return -IBGS_ERR_INVALID; }
    if (a.shs_rest && (!a.shs || a.M < 2 || (reinterpret_cast<uintptr_t>(a.shs_rest) & 15u) || (reinterpret_cast<uintptr_t>(a.shs) & 15u))) {
        set_error("shs_rest needs shs (the DC coefficients), M >= 2 and 16-byte aligned arrays"); return -IBGS_ERR_INVALID;
    }
    if (a.render_geo && a.render_depth_only) { set_error("render_geo together with render_depth_only is not supported"); return -IBGS_ERR_INVALID; }
    if (a.plane_mode != IBGS_PLANE_NONE) {
        if (a.all_map) { set_error("give either all_map or plane_mode, not both"); return -IBGS_ERR_INVALID; }
        if (a.plane_mode == IBGS_PLANE_LEARNT ? !a.plane_normal : (a.plane_mode != IBGS_PLANE_SMALLEST_AXIS || !a.scales || !a.rotations)) {
            set_error("plane_mode %d: plane_normal (learnt) or scales + rotations (smallest axis) required", a.plane_mode); return -IBGS_ERR_INVALID;
        }
    }
    if ((a.render_geo || a.render_depth_only) && !a.all_map && a.plane_mode == IBGS_PLANE_NONE) { set_error("all_map (or plane_mode) is required for render_geo / render_depth_only"); return -IBGS_ERR_INVALID; }
    if (a.render_geo || a.render_depth_only) {
        if (a.buffer_length < 1 || a.buffer_length > IBGS_MAX_BUFFER_LENGTH) { set_error("buffer_length %d outside 1..%d", a.buffer_length, IBGS_MAX_BUFFER_LENGTH); return -IBGS_ERR_INVALID; }
        if (!a.out_depth) { set_error("out_depth required"); return -IBGS_ERR_INVALID; }
    }
    if (a.render_geo) {
        if (a.n_src < 1 || a.n_src > IBGS_MAX_SRC) { set_error("n_src %d outside 1..%d", a.n_src, IBGS_MAX_SRC); return -IBGS_ERR_INVALID; }
        if (!a.ref_to_src || !a.src_cam_pos || !a.src_images || !a.src_depths) { set_error("source view pointers required for render_geo"); return -IBGS_ERR_INVALID; }
        if (a.flags & IBGS_FLAG_SRC_DEPTH_SLOTS) {
            for (int m = 0; m < a.n_src; m++) if (a.src_depth_slot[m] < 0) { set_error("src_depth_slot[%d] = %d is negative", m, a.src_depth_slot[m]); return -IBGS_ERR_INVALID; }
        }
        if (!a.out_normal || !a.out_cam_feat || !a.out_warped || !a.out_min_depth_diff || !a.out_camera_ray || !a.out_mask) { set_error("geo outputs required"); return -IBGS_ERR_INVALID; }
        if (!a.tex || a.tex_bytes < ibgs_required_tex(a.n_src, a.W, a.H)) { set_error("tex scratch too small"); return -IBGS_ERR_ALLOC; }
    }
    if (!a.render_depth_only && !a.out_color) { set_error("out_color required"); return -IBGS_ERR_INVALID; }
    // batched depth-only views: one tall tile grid of nv x ceil(H/16) rows, nv x P instances
    const int nv = a.n_views > 1 ? a.n_views : 1;
    if (nv > 1) {
        if (nv > IBGS_MAX_VIEWS || !a.render_depth_only || a.plane_mode == IBGS_PLANE_NONE) {
            set_error("n_views = %d needs render_depth_only, plane_mode != 0 and n_views <= %d", a.n_views, IBGS_MAX_VIEWS); return -IBGS_ERR_INVALID;
        }
        if ((int64_t)nv * a.P > 0x7FFFFFFF) { set_error("n_views x P too large"); return -IBGS_ERR_INVALID; }
    }
    const int Pn = nv * a.P;                                      // instances
    const int Hn = nv > 1 ? nv * TILE * ((a.H + TILE - 1) / TILE) : a.H;   // height of the stacked grid in pixels
    if (!a.geom || a.geom_bytes < ibgs_required_geom(Pn)) { set_error("geom arena too small"); return -IBGS_ERR_ALLOC; }
    if (!a.img || a.img_bytes < ibgs_required_img(a.W, Hn)) { set_error("img arena too small"); return -IBGS_ERR_ALLOC; }
    if (!a.binning_alloc) { set_error("binning_alloc callback required"); return -IBGS_ERR_INVALID; }

    int rc;
    GeomState g = GeomState::carve(a.geom, (size_t)Pn, nullptr);
    ImgState im = ImgState::carve(a.img, a.W, Hn, nullptr);
    const int gx = (a.W + TILE - 1) / TILE, gy = nv * ((a.H + TILE - 1) / TILE);

    const bool deferred = a.rendered_hint > 0 && !debug;
    const size_t nwaves = (size_t)nv * (((size_t)a.P + 63) / 64);          // words of per-wave tile sums the preprocess kernel wrote; the depth sort's error flag follows them
    RSlot* rs = rslot(s, true);
    if (!rs) return -IBGS_ERR_HIP;
    uint32_t ticket = 0;
    { StageTimer t(s, IBGS_STAGE_PREPROCESS);
      if ((rc = launch_preprocess(s, a, g, deferred ? 1 : 0))) return rc;
      if (deferred) {
          // R = the sum of the tiles touched, final as soon as the geometry kernel is: a one-workgroup kernel adds the per-wave sums up HERE and stores the
          // total + a ticket into pinned host memory, while the stream goes on with the SH colours and the depth sort.  (Until round 4 the sums left
          // behind the sort, by a copy on a second stream, together with the sort's error flag: on small frames the host then sat out five sort launches'
          // worth of GPU latency -- ~45 us of a 0.34 ms call pair, profiles/r05_host_split.txt -- before it could queue the loss and the backward.)
          ticket = ++rs->seq;          // (wraps after 4 G forwards on one stream: the comparison below is for equality)
          const RenderedNote note{g.tile_partial, (uint32_t)nwaves, ticket, rs->host};
          if ((rc = launch_preprocess(s, a, g, 2, &note)) < 0) return rc;          // (1: the SH colour kernel's first workgroup carries the note -- one launch fewer)
          if (rc == 0) { hipLaunchKernelGGL(rendered_note_kernel, dim3(1), dim3(1024), 0, s, note); IBGS_HIP(hipGetLastError()); }
      }
    }
    if ((rc = stage_check(s, debug, "preprocess"))) return rc;
    { StageTimer t(s, IBGS_STAGE_DEPTH_SORT);
      // (the preprocess kernel has zeroed the sort's scratch, its look-back error flag -- [Pn + 1], or the word behind the tile sums when R is
      // sized from a hint: that one travels back together with them --, and [Pn + 3], [Pn + 4])
      // Gaussians without tiles (key 0xFFFFFFFF) are not carried through the sort; [Pn + 3] = how many others there are
      if ((rc = radix_sort_pairs(s, g.sort_key, g.sort_val, (size_t)Pn, 32, g.hist, g.hist_elems, deferred ? g.tile_partial + nwaves : g.offsets + Pn + 1,
                                 g.offsets + Pn + 3, true, g.offsets + Pn + 4))) return rc; }
    if ((rc = stage_check(s, debug, "depth sort"))) return rc;
    // R = total number of (Gaussian, tile) pairs.  The binning arena is sized from it, and it is only known on the device.
    //  * no hint (first call of a shape, or debug): the tiles-touched counts are scanned, R travels to the host through a pinned
    //    word + event and the host waits for it here, as the reference does (rasterizer_impl.cu:430);
    //  * with args->rendered_hint the arena is carved for the hint at once and every remaining stage is enqueued; R = the sum of the tiles touched,
    //    which the preprocess kernel left as one partial sum per wave: those words and the depth sort's error flag are copied to the host right
    //    HERE, behind the sort, and the host adds them up only after everything is queued -- the GPU never idles on the round trip, no scan runs,
    //    and the host is released while binning, list scatter and render are still ahead of the GPU (when it waited for the binning's own count,
    //    a trained-opacity frame left it 0.13 ms to queue the loss and the backward, which takes it 0.16 ms: the GPU idled).
    { StageTimer t(s, IBGS_STAGE_SCAN);
      // (the total does not depend on the order: the per-Gaussian counts are scanned as they lie)
      if (!deferred && (rc = exclusive_scan_u32(s, g.tiles, g.offsets, (size_t)Pn, g.hist, g.hist_elems, true))) return rc; }
    auto exact_R = [&](int64_t* R_out) -> int {       // synchronous path: R from the scanned tile counts
        IBGS_HIP(hipMemcpyAsync(rs->host, g.offsets + Pn, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));      // R and the depth sort's error flag
        IBGS_HIP(hipEventRecord(rs->ev, s));
        IBGS_HIP(hipEventSynchronize(rs->ev));
        if (rs->host[1]) { set_error("depth sort: decoupled look-back timed out (lists would be mis-ordered)"); return -IBGS_ERR_HIP; }
        *R_out = (int64_t)rs->host[0];
        return 0;
    };
    int64_t R = 0, cap = 0;
    if (deferred) cap = a.rendered_hint < (int64_t)0xFFFF0000ll ? a.rendered_hint : (int64_t)0xFFFF0000ll;
    else { if ((rc = exact_R(&R))) return rc; cap = R; }

    auto tail = [&](int64_t n, bool read_back) -> int {
        int rc;
        // the hinted pass may use an arena the caller sized for the hint beforehand (ibgs_forward_args.binning: no call back into the caller, which
        // through ctypes costs ~10 us); anything else -- no hint, a too small hint -- asks binning_alloc
        const size_t need = ibgs_required_binning(n, a.W, Hn);
        char* bin_mem = (read_back && a.binning && a.binning_bytes >= need) ? a.binning : a.binning_alloc(need, a.binning_user);
        if (!bin_mem) { set_error("binning_alloc returned NULL for R=%lld", (long long)n); return -IBGS_ERR_ALLOC; }
        BinState b = BinState::carve(bin_mem, (size_t)n, a.W, Hn, nullptr);
        // hinted pass: the binning's last kernel leaves R as it counted it, the coarse slots in use and the depth sort's error word in pinned host memory
        // (diagnostics nobody waits for: ibgs_last_forward_stats; the error word: check_sort_flag)
        { StageTimer t(s, IBGS_STAGE_EMIT);
          if ((rc = launch_binning(s, Pn, n, gx, gy, g, b, im.ranges, a.tile_order_hint, im.meta, nv, read_back ? g.tile_partial + nwaves : nullptr, read_back ? rs->stats : nullptr, ticket))) return rc; }
        if (read_back) { rs->flag_pending = true; rs->last_stream = s; }
        { StageTimer t(s, IBGS_STAGE_TILE_SORT); if ((rc = launch_binning_scatter(s, n, gx, gy, b))) return rc; }
        if ((rc = stage_check(s, debug, "binning"))) return rc;
        const float4* rgba = nullptr;
        if (a.render_geo) {
            float4* t = reinterpret_cast<float4*>((reinterpret_cast<uintptr_t>(a.tex) + 127) & ~uintptr_t(127));
            if (!(a.flags & IBGS_FLAG_TEX_PACKED) && (rc = launch_pack_rgba(s, a.src_images, t, a.W, a.H, a.n_src))) return rc;
            rgba = t;
        }
        { StageTimer t(s, IBGS_STAGE_RENDER_FWD); if ((rc = launch_render_forward(s, a, g, b, im, rgba))) return rc; }
        return stage_check(s, debug, "render");
    };
    const bool prev_pending = rs->flag_pending;          // a hinted forward before this one whose sort nobody has vouched for yet
    if ((rc = tail(cap, deferred))) return rc;
    g_last_stats[1] = -1; g_last_stats[2] = 0;
    g_stats_slot = rs; rs->stats_pending = false;
    if (deferred) {
        {   // wait for the ticket: a few thousand polls of one pinned cache line, then yield between polls; bounded (a lost ticket must not hang the trainer)
            volatile uint32_t* tk = rs->host + 2;
            const uint32_t want = ticket;
            uint64_t spins = 0;
            struct timespec t0; bool timed = false;
            while (__atomic_load_n(const_cast<uint32_t*>(tk), __ATOMIC_ACQUIRE) != want) {
                if (++spins > 4000) {
                    if (!timed) { clock_gettime(CLOCK_MONOTONIC, &t0); timed = true; }
                    sched_yield();
                    if ((spins & 1023) == 0) {
                        struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
                        if (t1.tv_sec - t0.tv_sec > 5) {          // long wait: legitimate while the stream still has work in front of the note kernel
                            if (hipStreamQuery(s) == hipErrorNotReady) { t0 = t1; continue; }
                            IBGS_HIP(hipStreamSynchronize(s));          // the stream is empty (or broken): the ticket must be there now
                            if (*tk == want) break;
                            set_error("ibgs_forward: the device never reported R (ticket %u, last seen %u)", want, *tk); return -IBGS_ERR_HIP;
                        }
                    }
                }
            }
        }
        // (this stream has now passed every kernel of the forwards before this one: their sticky error word is final -- a forward no backward followed)
        if (prev_pending && *(volatile uint32_t*)(rs->stats + 3)) {
            rs->stats[3] = 0u;
            set_error("depth sort of the previous forward: decoupled look-back timed out (its lists were mis-ordered)"); return -IBGS_ERR_HIP;
        }
        const uint64_t sum = (uint64_t)rs->host[0] | ((uint64_t)rs->host[1] << 32);
        R = (int64_t)sum;          // exact, whatever the binning could fit (a coarse entry stands for at least one pair: C <= R, so R <= cap means nothing was dropped)
        if (R > cap) {
            g_last_stats[2] = 1;
            // The hint was too small: the lists above are truncated.  Drain the stream (the first arena may be released by the
            // second callback) and redo binning + render with the exact size (every output element is rewritten).  Same results
            // as without a hint, one wasted pass.
            IBGS_HIP(hipStreamSynchronize(s));
            if ((rc = check_sort_flag(rs, false))) return rc;          // (the stream is drained: this forward's own word is final too, and its note has arrived)
            if ((rc = tail(R, false))) return rc;
        } else rs->stats_pending = true;
    }
    g_last_stats[0] = R;
    return R;
}

int32_t ibgs_backward(const ibgs_backward_args* ap)
{
    if (!ap) { set_error("null args"); return -IBGS_ERR_INVALID; }
    const ibgs_backward_args& a = *ap;
    hipStream_t s = reinterpret_cast<hipStream_t>(a.stream);
    const bool debug = (a.flags & IBGS_FLAG_DEBUG) != 0;
    if (a.P <= 0) return 0;                                       // rasterize_points.cu:221
    RSlot* rs = rslot(s, false);          // the slot of the forward queued on this stream (whichever thread calls: PyTorch's autograd worker, usually)
    { int frc = check_sort_flag(rs, false); if (frc) return frc; }          // (no wait: reports the word if the GPU has already set it)
    if (!a.geom || !a.img || (!a.binning && a.R > 0)) { set_error("backward needs the forward's arenas"); return -IBGS_ERR_INVALID; }
    if (!a.grad_acc) { set_error("grad_acc scratch required"); return -IBGS_ERR_INVALID; }
    if (!a.dL_dmean2D || (!a.dL_dmean2D_abs && !(a.flags & IBGS_FLAG_NO_ABS_GRAD)) || !a.dL_dopacity || !a.dL_dmean3D) {
        set_error("missing gradient output"); return -IBGS_ERR_INVALID;
    }
    if (a.shs && !a.dL_dsh && !(a.flags & IBGS_FLAG_SH_FACTORED)) { set_error("dL_dsh required"); return -IBGS_ERR_INVALID; }
    if (a.shs_rest && (!a.shs || a.M < 2 || (!(a.flags & IBGS_FLAG_SH_FACTORED) && !a.dL_dsh_rest)
                       || ((reinterpret_cast<uintptr_t>(a.shs_rest) | reinterpret_cast<uintptr_t>(a.shs) | reinterpret_cast<uintptr_t>(a.dL_dsh_rest) | reinterpret_cast<uintptr_t>(a.dL_dsh)) & 15u))) {
        set_error("shs_rest needs shs, M >= 2, dL_dsh_rest and 16-byte aligned arrays"); return -IBGS_ERR_INVALID;
    }
    if (a.shs && a.M == 16 && ((reinterpret_cast<uintptr_t>(a.shs) & 15u) || (a.dL_dsh && (reinterpret_cast<uintptr_t>(a.dL_dsh) & 15u)))) {
        set_error("shs / dL_dsh (M = 16) must be 16-byte aligned"); return -IBGS_ERR_INVALID;
    }
    if (!a.dL_dcolors && (!a.shs || (a.flags & IBGS_FLAG_SH_FACTORED))) { set_error("dL_dcolors required (precomputed colours, or IBGS_FLAG_SH_FACTORED)"); return -IBGS_ERR_INVALID; }
    if (!a.dL_dcov3D && a.cov3D_precomp) { set_error("dL_dcov3D required with cov3D_precomp"); return -IBGS_ERR_INVALID; }
    if (a.scales && (!a.dL_dscale || !a.dL_drot)) { set_error("dL_dscale / dL_drot required"); return -IBGS_ERR_INVALID; }
    if (a.render_geo) {
        if (a.plane_mode == IBGS_PLANE_NONE ? (!a.all_map || !a.dL_dall_map)
                                            : (a.plane_mode == IBGS_PLANE_LEARNT ? (!a.plane_normal || !a.dL_dplane_normal)
                                                                                 : (a.plane_mode != IBGS_PLANE_SMALLEST_AXIS || !a.scales || !a.rotations))) {
            set_error("geo backward: plane inputs / gradient outputs missing"); return -IBGS_ERR_INVALID;
        }
        if (!a.out_depth || !a.out_warped || !a.ref_to_src || !a.src_images) { set_error("geo backward inputs missing"); return -IBGS_ERR_INVALID; }
        if (!a.tex || a.tex_bytes < ibgs_required_tex(a.n_src, a.W, a.H)) { set_error("tex scratch too small"); return -IBGS_ERR_ALLOC; }
        if (!a.geo_table || a.geo_table_bytes < (a.buffer_length > 0 ? ibgs_required_geo_table_for(a.W, a.H, a.buffer_length) : ibgs_required_geo_table(a.W, a.H))) { set_error("geo_table scratch too small"); return -IBGS_ERR_ALLOC; }
    }
    int rc;
    GeomState g = GeomState::carve(a.geom, (size_t)a.P, nullptr);
    ImgState im = ImgState::carve(a.img, a.W, a.H, nullptr);
    BinState b = BinState::carve(a.binning, (size_t)a.R, a.W, a.H, nullptr);
    const float4* rgba = nullptr;
    float* geo_tab = nullptr;
    if (a.render_geo) {
        float4* t = reinterpret_cast<float4*>((reinterpret_cast<uintptr_t>(a.tex) + 127) & ~uintptr_t(127));
        const bool window = a.dL_ddepth || a.dL_dwarped;          // only the window pass (median depth / warp gradients) reads the textures and fills the table
        if (window && !(a.flags & IBGS_FLAG_TEX_PACKED) && (rc = launch_pack_rgba(s, a.src_images, t, a.W, a.H, a.n_src))) return rc;
        rgba = t;
        geo_tab = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(a.geo_table) + 127) & ~uintptr_t(127));
    }
    if (a.R > 0) {
        const bool det = (a.flags & IBGS_FLAG_DETERMINISTIC) != 0;
        DetState ds{};
        int ipt = 1;
        if (det) {
            if (!a.det_scratch || a.det_scratch_bytes < ibgs_required_deterministic_for(a.R, a.P, a.W, a.H, a.render_geo, a.flags)) { set_error("det_scratch too small"); return -IBGS_ERR_ALLOC; }
            ipt = render_backward_waves_per_tile(a);
            ds = DetState::carve(a.det_scratch, (size_t)a.R * ipt, (size_t)a.P, nullptr);
            if ((rc = launch_det_prepare(s, ds, (size_t)a.R * ipt))) return rc;
        }
        // (launch_render_backward brackets its kernels itself: IBGS_STAGE_GEO_WINDOW, IBGS_STAGE_TILE_ORDER, IBGS_STAGE_RENDER_BWD = the blend kernel alone)
        if ((rc = launch_render_backward(s, a, g, b, im, rgba, det ? ds.slab : nullptr, geo_tab))) return rc;
        if (det) { StageTimer t(s, IBGS_STAGE_TILE_ORDER); if ((rc = launch_det_reduce(s, ds, b.point_list, (size_t)a.R, ipt, a.P, a.grad_acc, g.offsets + a.P))) return rc; }
        if ((rc = stage_check(s, debug, "render backward"))) return rc;
    }
    { StageTimer t(s, IBGS_STAGE_PREPROCESS_BWD); if ((rc = launch_preprocess_backward(s, a, g))) return rc; }
    if ((rc = stage_check(s, debug, "preprocess backward"))) return rc;
    // everything is queued: was the forward these gradients belong to sound?  (check_sort_flag: a bounded wait for its binning's note, which has normally long arrived)
    return check_sort_flag(rs, true);
}

int32_t ibgs_sh_grad_from_views(void* stream, int32_t P, int32_t D, int32_t M, int32_t n_views, const float* means3D,
                                const float* camposes, const float* dcolor, int64_t view_stride, float* dL_dsh)
{
    if (view_stride == 0) view_stride = (int64_t)P * 3;
    if (view_stride < (int64_t)P * 3) { set_error("view_stride smaller than P x 3"); return -IBGS_ERR_INVALID; }
    if (P <= 0 || M <= 0) return 0;
    if (D < 0 || D > 3 || (D + 1) * (D + 1) > M || n_views < 0) { set_error("bad SH degree / view count"); return -IBGS_ERR_INVALID; }
    if (!means3D || !dL_dsh || (n_views > 0 && (!camposes || !dcolor))) { set_error("null pointer"); return -IBGS_ERR_INVALID; }
    return launch_sh_grad_from_views(reinterpret_cast<hipStream_t>(stream), P, D, M, n_views, means3D, camposes, dcolor, (size_t)view_stride, dL_dsh);
}

int32_t ibgs_mark_visible(void* stream, int32_t P, const float* means3D, const float* viewmatrix,
                          const float* projmatrix, uint8_t* present)
{
    (void)projmatrix;
    if (P <= 0) return 0;
    if (!means3D || !viewmatrix || !present) { set_error("null pointer"); return -IBGS_ERR_INVALID; }
    return launch_mark_visible(reinterpret_cast<hipStream_t>(stream), P, means3D, viewmatrix, present);
}

}  // extern "C"
