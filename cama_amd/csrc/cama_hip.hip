// libcama_hip.so -- hand-written gfx950 (MI355X / CDNA4) kernels for the CAMA multi-camera
// reprojection hot path, behind the C ABI declared in include/cama_hip.h.
//
// Path (reference, /root/reference): cama/dataset.py:78-126 (yield_frame -> project_all_camera ->
// render_vectors), cama/reproject.py:108-131,187-205,246-257, cama/tools.py:22-25.
//
// Design (DESIGN.md has the full account):
//   * The work is pointwise fp64 geometry + byte streaming: HBM-bound, no MFMA.
//   * k_frames<MODE>: one thread per (frame, vertex).  The frame's world->chassis 3x4 and all
//     cameras' chassis->camera 3x4 + K 3x3 are staged once per workgroup in LDS (<= 2.8 KB);
//     the fp32 SoA vertex buffer is read once per frame for ALL cameras with fully coalesced
//     dword loads; arithmetic is fp64 k-ordered FMA chains (bit-identical to the reference's
//     numpy/OpenBLAS matmul), compiled with -ffp-contract=off so nothing else is fused.
//     MODE_EMIT writes (v,u)+visibility; MODE_COUNT / MODE_FILL bin "stamps" (one per visible
//     point per camera) by (frame, camera, row band) with wave-aggregated atomics.
//   * k_overlay: one workgroup per (frame, camera, band of R full image rows).  The band is a
//     single contiguous byte range of the source frame; it is read once with 16-byte loads and
//     written once, already at its 2x3-mosaic address.  Bands that received stamps first resolve
//     "last writer wins" deterministically: a per-pixel u32 owner table in LDS takes
//     atomicMax(draw index) over the stamps' disc footprints, then the copy patches bytes whose
//     pixel has an owner.  No pixel is read or written twice.
//   * 64-wide wavefronts throughout (ballots are 64-bit, scans step to 32).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <utility>
#include <vector>
#include <mutex>
#include <atomic>
#include <string>

#include "cama_internal.hpp"     // + cama_common.hpp, cama_hip_diag.h (which includes cama_hip.h: the contract)

// thread-local message of the most recent failure (cama_last_error); shared by the library's translation units (cama_common.hpp)
static thread_local char g_err[512] = "";
int cama_impl::fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

cama_impl::LaunchMods &cama_impl::launch_mods()
{
    static thread_local LaunchMods m;
    return m;
}

namespace {
using namespace cama_impl;          // the host structs (BinPlan, ScratchLayout, ScratchRef, BinCall, RawSource), fail(), align_up()
thread_local bool g_overlay_probe = false;     // the next plain overlay launch is cama_overlay_probe's: k_overlay_probe, contiguous order

// Opt-in timing of the dominant kernel (k_overlay) with HIP events recorded on the launch stream.
struct ProfileState {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;   // recorded, not yet collected (overlay)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_project;   // the same for k_frames_project*
    std::vector<hipEvent_t> pool;
};
thread_local ProfileState g_prof;

hipEvent_t prof_event()
{
    hipEvent_t e = nullptr;
    if (!g_prof.pool.empty()) {
        e = g_prof.pool.back();
        g_prof.pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
        e = nullptr;
    }
    return e;
}

#ifndef RAW35_PAIR_DEFAULT
#define RAW35_PAIR_DEFAULT 0      // k_overlay_raw35: bands per workgroup - 1 (see raw35_kernels.hpp)
#endif
constexpr int BLOCK = 256;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // one 16-byte chunk (global_load_dwordx4)
#ifndef OVERLAY_BLOCK
#define OVERLAY_BLOCK 256         // threads per overlay workgroup (one workgroup per band)
#endif
#ifndef OVERLAY_UNROLL
#define OVERLAY_UNROLL 5          // 16-byte chunks in flight per thread in the overlay copy loop
#endif
#ifdef OVERLAY_PLAIN_MEM
#define OVERLAY_LOAD(p) (*(p))
#define OVERLAY_STORE(v, p) (*(p) = (v))
#else                             // streaming data, never re-read: non-temporal hint
#define OVERLAY_LOAD(p) __builtin_nontemporal_load(p)
#define OVERLAY_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif
constexpr int CAM_STRIDE = 21;  // 12 (3x4 of chassis->camera) + 9 (K)

struct Crop { double v[6]; };
// Filled-circle footprint: row |dy| = k spans u +- hw(k).  Packed (4 bits per row + a row mask) so that the kernels
// read it from SGPRs; as an int array in the kernel arguments every dynamically indexed hw[k] was a global load (+ a
// full vmcnt drain) inside the rasterisation loop.
struct Disc { int radius; uint32_t rows; uint64_t hw4; };
__device__ __forceinline__ int disc_halfwidth(const Disc &d, int k)
{
    return ((d.rows >> k) & 1u) ? (int)((d.hw4 >> (4 * k)) & 15ull) : -1;
}
struct Palette { uint32_t c[2]; uint32_t alpha256; };  // colours b | g<<8 | r<<16; alpha in 1/256 (256 = opaque)

#include "project_kernels.hpp"
#include "remap_device.hpp"
#include "overlay_kernels.hpp"
#include "raw35_kernels.hpp"
#include "resample_kernels.hpp"
#include "map_kernels.hpp"
#include "egress_kernels.hpp"

// ------------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------------
int make_disc(int radius, const int32_t *hw, Disc &d)
{
    if (radius < 0 || radius > CAMA_MAX_RADIUS || !hw) return -1;
    d.radius = radius;
    d.rows = 0;
    d.hw4 = 0;
    for (int k = 0; k <= radius; ++k) {
        if (hw[k] > radius) return -1;           // owner rows carry `radius` spare cells per side (rasterise_one_padded)
        if (hw[k] < 0) continue;                 // row not drawn
        d.rows |= 1u << k;
        d.hw4 |= (uint64_t)hw[k] << (4 * k);
    }
    return 0;
}

Palette make_palette(const uint8_t *bgr, uint32_t alpha256 = 256u)
{
    Palette p;
    p.alpha256 = alpha256;
    for (int k = 0; k < 2; ++k)
        p.c[k] = (uint32_t)bgr[3 * k] | ((uint32_t)bgr[3 * k + 1] << 8) | ((uint32_t)bgr[3 * k + 2] << 16);
    return p;
}

}  // namespace
int cama_impl::band_rows_for(int W)
{
    // Measured on MI355X (profiles/): ~20 KB of image per workgroup streams best (more, smaller
    // workgroups balance stamped bands and keep the LDS owner table R*W*4 <= 26 KB -> 6 workgroups per CU).
    // R must stay >= 2*radius so a disc touches at most two bands.
    if (W >= 600) return 4;
    if (W >= 300) return 8;
    return 16;
}
namespace {

int log2i(int v)
{
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}


// Scratch of one launch = two parts (round 4):
//   plan part   what the cull pre-pass writes and the chain reads: work-list / candidate counters, the camera masks, the work
//               and candidate lists, the per-frame item counters and ranks.  Its size depends on (N, F, C) only.
//   stamp part  band counters, segment counts, the per-wave compacted stamps (stamps0) and the band-sorted stamps.  Sized for
//               the worst case -- every vertex visible in every camera: 24 B per (frame, camera, vertex) -- unless the launch
//               was PLANNED: for site-sized maps the pre-pass knows, before the projection runs, how many (block, frame) items
//               survive per frame and how many (wave, camera) chains they carry, which bounds both buffers exactly (BinPlan).
// Callers that bring their own scratch (cama_render_frames & co.) get both parts in one buffer, plan part first, worst-case
// sized.  A pipeline that owns its scratch (cama_pipeline_render* with scratch0 == NULL) keeps them apart and plans.
}  // namespace
int cama_impl::layout_scratch(int64_t N, int F, int C, int H, int W, int radius, ScratchLayout &L, const BinPlan *plan)
{
    L.R = plan && plan->band_rows > 0 ? plan->band_rows : band_rows_for(W);
    L.NB = (H + L.R - 1) / L.R;
    L.bands_per_stamp = radius > 0 ? 2 : 1;  // 2r+1 rows touch <= 2 bands when 2r <= R (checked by the caller)
    const size_t nbins = (size_t)F * C * L.NB, nfc = (size_t)F * C;
    L.planned = plan && plan->planned;
    L.segments = plan && plan->segments;
    L.wu = L.segments && plan->wu;
    L.record_bytes = L.segments ? 16 : 8;
    L.capacity = L.planned ? plan->capacity : (uint64_t)F * C * (uint64_t)N * L.bands_per_stamp;
    if (L.segments) L.capacity = plan->have_sorted_capacity ? plan->sorted_capacity : 0;   // (its own buffer: ScratchRef::sorted)
    // one segment per wave of the projection grid; planned: per wave of a SURVIVING block of the frame (rank order)
    L.nseg = L.planned ? plan->nseg : (uint32_t)((N + BLOCK - 1) / BLOCK) * (BLOCK / SEG);
    // ---- plan part
    // work lists of the crop cull: 8 (one per XCD), each up to ceil(vblocks / 8) * F uint32 items
    L.list_cap = (size_t)(((N + BLOCK - 1) / BLOCK + 7) / 8) * (size_t)F;
    size_t off = 0;
    L.work_count = off; off += 256;            // [0..7] work-list lengths, [8..15] candidate-list lengths, [16] "candidates ran"
    L.frame_items = off; off = align_up(off + (size_t)F * 4, 256);     // surviving blocks per frame (rank counters)
    L.demand = off; off += 256;                // [0] u64 (wave, camera) chains of the surviving blocks
    L.plan_zero_bytes = off;                   // work_count .. demand: one memset before the pre-pass
    L.cam_mask = off; off = align_up(off + (size_t)F * ((N + BLOCK - 1) / BLOCK) * 8, 256);   // per (frame, vertex block): 4 x u16
    L.cam_fn = off; off = align_up(off + (size_t)CAMA_MAX_CAMERAS * 20 * 8, 256);
    L.work = off;     off = align_up(off + 8 * L.list_cap * 4, 256);
    L.work_rank = off; off = align_up(off + 8 * L.list_cap * 4, 256);  // rank of a listed block among its frame's survivors
    // candidate pre-pass (k_block_candidates): per-frame world-space crop AABBs and the 8 (block, frame) candidate lists
    L.frame_box = off; off = align_up(off + (size_t)F * 6 * 8, 256);
    L.cand = off;     off = align_up(off + 8 * L.list_cap * 4, 256);             // 8 lists, like the work lists
    L.plan_total = off;
    // ---- stamp part
    off = 0;
    L.counts = off;   off = align_up(off + nbins * 4, 256);
    L.cursor = off;   off = align_up(off + nbins * 4, 256);
    L.seg_cnt = off;  off = align_up(off + nfc * L.nseg, 256);      // one byte per (frame, camera, segment)
    L.zero_bytes = off - L.counts;
    L.bin_off = off;  off = align_up(off + nbins * 4, 256);
    L.fc_total = off; off = align_up(off + nfc * 4, 256);
    L.fc_base = off;  off = align_up(off + nfc * 4, 256);
    L.stamps0 = off;  off = align_up(off + nfc * L.nseg * SEG * L.record_bytes + 16, 256);   // compacted per-segment stamps
    if (L.segments) {
        L.stamps = 0;                                                            // offset inside the sorted buffer
    } else {
        L.stamps = off;   off = align_up(off + (size_t)L.capacity * 8 + 8, 256);   // >= 1 record: empty bins read stamps[0]
    }
    L.stamp_total = off;
    L.total = L.plan_total + L.stamp_total;
    return 0;
}
namespace {

// a caller's single buffer: plan part first, worst-case sized (no plan)
}  // namespace
cama_impl::ScratchRef cama_impl::legacy_scratch(const void *scratch, size_t scratch_bytes, int64_t N, int F, int C, int H, int W, int radius)
{
    ScratchRef r;
    if (!scratch || N < 0 || F < 0 || C < 1 || H < 1 || W < 1 || radius < 0) return r;
    ScratchLayout L;
    layout_scratch(N, F, C, H, W, radius, L);
    r.plan = (char *)scratch;
    r.plan_bytes = std::min(scratch_bytes, L.plan_total);
    r.stamp = (char *)scratch + L.plan_total;
    r.stamp_bytes = scratch_bytes > L.plan_total ? scratch_bytes - L.plan_total : 0;
    return r;
}
namespace {

constexpr int CAND_COUNT_WORD = 8;     // work_count[0..7] = the 8 work lists' lengths, [8..15] the candidate lists', [16] =
                                       // "the candidate pre-pass ran" (cama_bin_stats)

// (block, frame) items from which the crop cull goes through a work list + persistent workgroups; and how many of those
// workgroups (256 CUs x 8 resident).  Env overrides are for A/B measurements only.
bool test_hook(const char *name, int64_t *value);
uint64_t cull_list_threshold();
// vertex blocks one projection workgroup runs: 1 until the launch has >= 16 k (block, frame) items, then as many as keep
// ~16 k workgroups (64 per CU), at most 8.  (test hook project_vb=n overrides)
int project_blocks_per_workgroup(uint64_t items)
{
    static const int forced = [] { int64_t v = 0; (void)test_hook("project_vb", &v); return (int)v; }();   // TEST HOOK (fuzz: 3 / 8)
    if (forced > 0) return forced;
    const uint64_t vb = items / 16384ull;
    return vb < 1 ? 1 : (vb > 8 ? 8 : (int)vb);
}

constexpr unsigned persistent_workgroups() { return 2048u; }      // 256 CUs x 8 resident

// Workgroup -> band mapping of the overlay kernels (overlay_kernels.hpp: xcd_item_of): 0 = workgroup L renders
// band L (every XCD owns every eighth band of one stream), 31 = every XCD renders one contiguous eighth of the launch.
// Chosen by the bytes a launch touches (frames read + mosaic written): measured on MI355X, kernel bandwidth / 8 TB/s,
// interleaved vs contiguous --
//   1600x900:  10 frames (0.5 GB) 0.79 / 0.76   20 (1 GB) 0.78 / 0.77   40 (2.1 GB) 0.755 / 0.815   160 (8 GB) 0.62 / 0.83
//              640 (33 GB) 0.66 / 0.80          960x540:  40 (0.75 GB) 0.77 / 0.72   80 (1.5 GB) 0.78 / 0.75   160 (3 GB) 0.76 / 0.76
// -- interleaved is the better order while every XCD can keep the whole launch's pages mapped; beyond ~2 GB per launch an
// XCD that touches only its own eighth of the pages wins, and the more the longer the launch.
// Round 3, later: both figures of a column were taken in DIFFERENT processes, and the contiguous order turned out to run in
// one of three speed modes that is fixed per process (profiles/r03_process_modes.txt: 0.75 / 0.79 / 0.82 at 40 frames, the
// interleaved order 0.755 in all of them).  Measured inside single processes instead (whole step / 8 TB/s, 8 processes):
// contiguous 0.70 .. 0.79, interleaved 0.73-0.74 in every one, round-robin chunks of 32 bands 0.74-0.76 in every one.  So:
//   * launches below 1.75 GiB, and every launch of the raw-frame / translucent variants: chunks of 32 bands (>= the
//     interleaved order wherever measured: 960x540 raw +1 %, 20 frames of 1600x900 +0.7 %; the contiguous order at 1.41 GB
//     of raw frames: 0.63-0.71 against 0.72 in every process);
//   * launches of 1.75 GiB and more: the process finds out which of {contiguous, chunks of 32} is faster HERE -- the first
//     launches alternate between the two with their own start / stop events, and once each has two timings the faster
//     one (per byte) is kept for the life of the process (MapTuner below; cama_overlay_mapping_info() reports it).
// The mapping is a bijection either way: pixels never depend on it.
// Option overlay_chunk_log2 = 0 .. 31 forces one mapping (no tuning).
constexpr uint32_t MAP_CHUNKED = 5u, MAP_CONTIGUOUS = 31u;
constexpr size_t MAP_BIG_LAUNCH = (size_t)7 << 28;               // 1.75 GiB

// TEST HOOKS: ONE environment variable, CAMA_TEST_HOOKS = "name[=value],name[=value],..." (read once per process), lets the
// parity suite force every production code path in child processes -- the options below by name, and the binning chain's
// switches (no_cam_mask, no_candidates, no_plan, project_vb=n).  None can change a result.  (Until round 5 each was its own
// environment variable, and the options had one each as well: thirteen in all.)
bool test_hook(const char *name, int64_t *value)
{
    static const std::vector<std::pair<std::string, int64_t>> hooks = [] {
        std::vector<std::pair<std::string, int64_t>> h;
        const char *e = getenv("CAMA_TEST_HOOKS");
        if (!e) return h;
        std::string s(e);
        size_t at = 0;
        while (at <= s.size()) {
            const size_t end = std::min(s.find(',', at), s.size());
            const std::string item = s.substr(at, end - at);
            const size_t eq = item.find('=');
            if (!item.empty())
                h.emplace_back(item.substr(0, eq), eq == std::string::npos ? 1 : strtoll(item.c_str() + eq + 1, nullptr, 10));
            at = end + 1;
        }
        return h;
    }();
    for (const auto &kv : hooks)
        if (kv.first == name) {
            if (value) *value = kv.second;
            return true;
        }
    return false;
}

// Process-wide options (cama_set_option / cama_get_option, include/cama_hip_diag.h): schedules and orders only -- none of them
// can change a result.  Changed at run time through the API (the parity suite forces each); no environment variables.
// Retired in round 6, their A/Bs settled: overlay_tune (the per-pair order trial stays on), bin_priority (highest: 0 / 1 / 2
// measure the same, profiles/r06_raw35_knobs.txt), pipeline_depth (two slots: three measured the same everywhere).
struct Option { const char *name; int64_t fallback; std::atomic<int64_t> value; std::atomic<bool> loaded; };
Option g_options[] = {
    {"overlay_chunk_log2", -1, {0}, {false}},   // -1 = library's choice, 0..31 = forced workgroup -> band order
    {"cull_list_min", 16384, {0}, {false}},     // (block, frame) items from which a site-sized map's cull goes through work lists
    {"pipeline_host_wait", -1, {0}, {false}},   // 1: cama_pipeline_* wait for a launch's binning on the HOST before queueing its
                                                // overlay (no barrier packet between consecutive overlays; the call blocks
                                                // ~0.1 ms); 0: stream-side wait; -1: host wait for plain launches that move >= 512 MiB
    {"band_rows", 0, {0}, {false}},             // rows per band of a pipeline's plain launches: 0 = per launch from the map's
                                                // measured stamp density, 4 | 8 = forced
};
static_assert(sizeof(g_options) / sizeof(g_options[0]) == OPT_COUNT, "option table");
}  // namespace
int64_t cama_impl::option(int k)
{
    Option &o = g_options[k];
    if (!o.loaded.load(std::memory_order_acquire)) {
        int64_t v = o.fallback;
        (void)test_hook(o.name, &v);
        o.value.store(v, std::memory_order_relaxed);
        o.loaded.store(true, std::memory_order_release);
    }
    return o.value.load(std::memory_order_relaxed);
}
namespace {
uint64_t cull_list_threshold() { return (uint64_t)std::max<int64_t>(option(OPT_CULL_LIST_MIN), 0); }
int overlay_forced_chunk_log2()
{
    const int64_t forced = option(OPT_CHUNK_LOG2);
    return forced > 31 ? 31 : (int)forced;
}
// the order of a launch that does not go through the tuner (small launches; the raw-frame and translucent variants): chunks
uint32_t overlay_chunk_log2()
{
    const int forced = overlay_forced_chunk_log2();
    return forced >= 0 ? (uint32_t)forced : MAP_CHUNKED;
}

// Which mapping does a big launch get?  Round 4: the speed of the contiguous order is a property of the BUFFERS a launch walks
// (their physical placement), not of the process -- one process that cycles through 12 (frames, mosaic) pairs sees 0.313 ..
// 0.350 ms on them, each pair always the same (profiles/r04_overlay_modes.txt) -- so the choice is kept per (device, frames
// pointer, mosaic pointer, bytes): the first launches over a pair alternate between the two orders with their own start /
// stop events, and once each has three timings the faster MEDIAN (per byte) is kept for that pair.  A table of the 64 most
// recently used pairs; a pair that is never seen six times (a stream of new buffers) simply keeps the contiguous order, which
// is the better one on average for launches of this size (40 frames: 0.75 .. 0.835 against 0.775 .. 0.79 for the chunks;
// 167 frames: 0.765 against 0.735).  Process-wide state behind a mutex (include/cama_hip.h, "global state").
struct MapTrial { uint32_t chunk_log2; hipEvent_t e0, e1; int which; uint64_t key_id; };     // e0 == nullptr: not a trial
struct MapTuner {
    static constexpr int SAMPLES = 3, MAX_KEYS = 64;
    std::mutex mu;
    struct Entry {
        int device; const void *src, *dst; size_t bytes;
        uint64_t id, stamp;
        int decided = -1;                     // -1 = still measuring, else MAP_CONTIGUOUS | MAP_CHUNKED
        int issued = 0;
        int done[2] = {0, 0};
        double t[2][SAMPLES] = {};            // seconds per byte: [0] contiguous, [1] chunked
    };
    std::vector<Entry> entries;
    uint64_t clock = 0, next_id = 1, last_id = 0;
    struct Pending { hipEvent_t e0, e1; int which; double bytes; uint64_t key_id; };
    std::vector<Pending> pending;

    Entry *find(uint64_t id)
    {
        for (auto &e : entries)
            if (e.id == id) return &e;
        return nullptr;
    }
    static double median(const double *v, int n)
    {
        double a[SAMPLES];
        std::copy(v, v + n, a);
        std::sort(a, a + n);
        return n & 1 ? a[n / 2] : 0.5 * (a[n / 2 - 1] + a[n / 2]);
    }
    void poll()
    {
        for (size_t k = 0; k < pending.size();) {
            const hipError_t q = hipEventQuery(pending[k].e1);
            if (q == hipErrorNotReady) { ++k; continue; }
            float ms = 0.f;
            Entry *e = find(pending[k].key_id);
            if (q == hipSuccess && e && hipEventElapsedTime(&ms, pending[k].e0, pending[k].e1) == hipSuccess && ms > 0.f &&
                e->done[pending[k].which] < SAMPLES)
                e->t[pending[k].which][e->done[pending[k].which]++] = (double)ms * 1e-3 / pending[k].bytes;
            (void)hipEventDestroy(pending[k].e0);         // (also when the launch or the query failed: nothing leaks)
            (void)hipEventDestroy(pending[k].e1);
            pending[k] = pending.back();
            pending.pop_back();
            if (e && e->decided < 0 && e->done[0] >= SAMPLES && e->done[1] >= SAMPLES)
                e->decided = median(e->t[1], SAMPLES) < median(e->t[0], SAMPLES) * 0.995 ? (int)MAP_CHUNKED : (int)MAP_CONTIGUOUS;
        }
        (void)hipGetLastError();              // (hipEventQuery's "not ready" is not an error of ours)
    }
    // the mapping for a big launch over (src, dst); a trial if the caller can give the launch its own start / stop events
    MapTrial pick(const void *src, const void *dst, size_t bytes, bool can_trial)
    {
        int device = 0;
        (void)hipGetDevice(&device);
        std::lock_guard<std::mutex> lock(mu);
        poll();
        Entry *e = nullptr;
        for (auto &c : entries)
            if (c.device == device && c.src == src && c.dst == dst && c.bytes == bytes) { e = &c; break; }
        if (!e) {
            if ((int)entries.size() >= MAX_KEYS) {        // evict the least recently used pair
                size_t lru = 0;
                for (size_t k = 1; k < entries.size(); ++k)
                    if (entries[k].stamp < entries[lru].stamp) lru = k;
                entries[lru] = entries.back();
                entries.pop_back();
            }
            entries.emplace_back();
            e = &entries.back();
            e->device = device; e->src = src; e->dst = dst; e->bytes = bytes; e->id = next_id++;
        }
        e->stamp = ++clock;
        last_id = e->id;
        if (e->decided >= 0) return MapTrial{(uint32_t)e->decided, nullptr, nullptr, 0, e->id};
        // a pair whose trials never completed (timing failures) stays on the order whose speed does not depend on the pair;
        // a launch that merely cannot carry a trial right now (it is being profiled) follows the contiguous order
        if (e->issued >= 4 * SAMPLES) return MapTrial{MAP_CHUNKED, nullptr, nullptr, 0, e->id};
        if (!can_trial) return MapTrial{MAP_CONTIGUOUS, nullptr, nullptr, 0, e->id};
        MapTrial t{MAP_CONTIGUOUS, nullptr, nullptr, e->issued & 1, e->id};
        if (hipEventCreate(&t.e0) != hipSuccess || hipEventCreate(&t.e1) != hipSuccess) {
            if (t.e0) (void)hipEventDestroy(t.e0);
            return MapTrial{MAP_CONTIGUOUS, nullptr, nullptr, 0, e->id};
        }
        t.chunk_log2 = t.which ? MAP_CHUNKED : MAP_CONTIGUOUS;
        ++e->issued;
        return t;
    }
    void submitted(const MapTrial &t, double bytes)
    {
        std::lock_guard<std::mutex> lock(mu);
        pending.push_back(Pending{t.e0, t.e1, t.which, bytes, t.key_id});
    }
    void abandon(const MapTrial &t)           // a trial whose launch did not happen
    {
        if (t.e0) (void)hipEventDestroy(t.e0);
        if (t.e1) (void)hipEventDestroy(t.e1);
    }
};
MapTuner g_map_tuner;

dim3 overlay_grid(size_t items, uint32_t chunk_log2)
{
    if (chunk_log2 >= 31u) return dim3((unsigned)((items + 7) / 8 * 8));
    const size_t per = (size_t)8 << chunk_log2;
    return dim3((unsigned)((items + per - 1) / per * per));
}

}  // namespace
int cama_impl::check_common(int64_t N, int F, int C, int W, int H)
{
    if (N < 0 || N >= (1ll << 30)) return fail(CAMA_EINVAL, "N=%lld out of range [0, 2^30)", (long long)N);
    if (F < 0 || F > 65535) return fail(CAMA_EINVAL, "F=%d out of range [0, 65535]", F);
    if (C < 1 || C > CAMA_MAX_CAMERAS) return fail(CAMA_EINVAL, "C=%d out of range [1, %d]", C, CAMA_MAX_CAMERAS);
    if (W < 1 || H < 1 || W > 65535 || H > 65535) return fail(CAMA_EINVAL, "W x H = %d x %d out of range", W, H);
    return 0;
}
namespace {

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int cama_abi_version(void) { return CAMA_ABI_VERSION; }
const char *cama_last_error(void) { return g_err; }

int cama_circle_halfwidths(int32_t radius, int32_t *hw)
{
    if (radius < 0 || radius > CAMA_MAX_RADIUS || !hw) return fail(CAMA_EINVAL, "radius %d out of range", radius);
    // OpenCV imgproc drawing.cpp Circle(), fill branch: union of the horizontal spans it draws per row
    for (int k = 0; k <= radius; ++k) hw[k] = -1;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        if (dx > hw[dy]) hw[dy] = dx;
        if (dy > hw[dx]) hw[dx] = dy;
        ++dy;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
    return radius + 1;
}

int cama_overlay_band_rows(int32_t W) { return band_rows_for(W); }

int cama_profile_enable(int32_t on)
{
    g_prof.on = on != 0;
    return CAMA_OK;
}

static int profile_drain(std::vector<std::pair<hipEvent_t, hipEvent_t>> &pending, double *total_ms, int32_t *launches)
{
    double sum = 0.0;
    int n = 0;
    for (auto &pr : pending) {
        HIP_TRY(hipEventSynchronize(pr.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        sum += ms;
        ++n;
        g_prof.pool.push_back(pr.first);
        g_prof.pool.push_back(pr.second);
    }
    pending.clear();
    if (total_ms) *total_ms = sum;
    if (launches) *launches = n;
    return CAMA_OK;
}

int cama_profile_collect(double *total_ms, int32_t *launches) { return profile_drain(g_prof.pending, total_ms, launches); }

// per-launch durations of the timed overlay launches, in issue order: up to `capacity` values into ms[], *launches = how many
// were pending (all of them are drained)
int cama_profile_collect_each(double *ms, int32_t capacity, int32_t *launches)
{
    if ((!ms && capacity > 0) || capacity < 0) return fail(CAMA_EINVAL, "bad arguments");
    int n = 0;
    for (auto &pr : g_prof.pending) {
        HIP_TRY(hipEventSynchronize(pr.second));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, pr.first, pr.second));
        if (n < capacity) ms[n] = t;
        ++n;
        g_prof.pool.push_back(pr.first);
        g_prof.pool.push_back(pr.second);
    }
    g_prof.pending.clear();
    if (launches) *launches = n;
    return CAMA_OK;
}
int cama_profile_collect_project(double *total_ms, int32_t *launches)
{
    return profile_drain(g_prof.pending_project, total_ms, launches);
}

int cama_transform_points(const void *xyz, int32_t xyz_is_f64, int64_t N, const double *T, int32_t F,
                          const double *crop, double *out_xyz, uint8_t *crop_mask, void *stream)
{
    if (int rc = check_common(N, F, 1, 1, 1)) return rc;
    if (!xyz && N) return fail(CAMA_EINVAL, "xyz is NULL");
    if (!T && F) return fail(CAMA_EINVAL, "T is NULL");
    if (N == 0 || F == 0) return CAMA_OK;
    Crop c{};
    if (crop) memcpy(c.v, crop, sizeof(c.v));
    dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK), (unsigned)F);
    hipStream_t s = (hipStream_t)stream;
    if (xyz_is_f64)
        hipLaunchKernelGGL(k_transform_points<double>, grid, dim3(BLOCK), 0, s, (const double *)xyz, N, T, c,
                           crop ? 1 : 0, out_xyz, crop_mask);
    else
        hipLaunchKernelGGL(k_transform_points<float>, grid, dim3(BLOCK), 0, s, (const float *)xyz, N, T, c,
                           crop ? 1 : 0, out_xyz, crop_mask);
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

int cama_crop_points(const double *xyz, int64_t n, const double *crop, uint8_t *mask, void *stream)
{
    if (int rc = check_common(n, 1, 1, 1, 1)) return rc;
    if (n == 0) return CAMA_OK;
    if (!xyz || !crop || !mask) return fail(CAMA_EINVAL, "NULL pointer argument");
    Crop c;
    memcpy(c.v, crop, sizeof(c.v));
    hipLaunchKernelGGL(k_crop_points, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                       xyz, n, c, mask);
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

int cama_project_points(const double *chassis_xyz, int64_t n, const double *c2cam, const double *K, int32_t C,
                        int32_t W, int32_t H, double *vu, uint8_t *vis, void *stream)
{
    if (int rc = check_common(n, 1, C, W, H)) return rc;
    if (n == 0) return CAMA_OK;
    if (!chassis_xyz || !c2cam || !K || !vu || !vis) return fail(CAMA_EINVAL, "NULL pointer argument");
    hipLaunchKernelGGL(k_project_points, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0,
                       (hipStream_t)stream, chassis_xyz, n, c2cam, K, C, W, H, vu, vis);
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

int cama_project_frames(const void *x, const void *y, const void *z, int32_t xyz_is_f64, int64_t N, const double *w2c,
                        int32_t F,
                        const double *c2cam, const double *K, int32_t C, const double *crop, int32_t W, int32_t H,
                        double *vu, uint8_t *vis, uint8_t *crop_mask, void *stream)
{
    if (int rc = check_common(N, F, C, W, H)) return rc;
    if (N == 0 || F == 0) return CAMA_OK;
    if (!x || !y || !z || !w2c || !c2cam || !K || !crop || !vu || !vis)
        return fail(CAMA_EINVAL, "NULL pointer argument");
    FrameArgs a{};
    a.x = x; a.y = y; a.z = z; a.N = N;
    a.w2c = w2c; a.c2cam = c2cam; a.K = K; a.C = C; a.W = W; a.H = H;
    memcpy(a.crop.v, crop, sizeof(a.crop.v));
    a.vu = vu; a.vis = vis; a.crop_mask = crop_mask;
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK), (unsigned)F);
    if (xyz_is_f64)
        hipLaunchKernelGGL(k_frames_emit<double>, grid, dim3(BLOCK), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(k_frames_emit<float>, grid, dim3(BLOCK), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

size_t cama_render_scratch_bytes(int64_t N, int32_t F, int32_t C, int32_t H, int32_t W, int32_t radius)
{
    if (N < 0 || F < 0 || C < 1 || H < 1 || W < 1 || radius < 0) return 0;
    ScratchLayout L;
    layout_scratch(N, F, C, H, W, radius, L);
    return L.total;
}

// validation shared by the bin / overlay halves of the fused render
extern "C++" int cama_impl::check_render(int64_t N, int32_t F, int32_t C, int32_t W, int32_t H, int32_t radius, const ScratchRef &sc,
                        ScratchLayout &L)
{
    if (int rc = check_common(N, F, C, W, H)) return rc;
    if (radius < 0 || radius > CAMA_MAX_RADIUS) return fail(CAMA_EINVAL, "radius %d out of range", radius);
    if (!sc.plan || !sc.stamp) return fail(CAMA_EINVAL, "scratch is NULL");
    layout_scratch(N, F, C, H, W, radius, L, &sc.bin_plan);
    if (sc.plan_bytes < L.plan_total || sc.stamp_bytes < L.stamp_total)
        return fail(CAMA_EINVAL, "scratch too small: %zu < %zu", sc.plan_bytes + sc.stamp_bytes, L.total);
    if (L.segments && sc.bin_plan.have_sorted_capacity && (!sc.sorted || sc.sorted_bytes < (size_t)L.capacity * 16 + 16))
        return fail(CAMA_EINVAL, "segment records: the sorted buffer is too small");
    if (L.capacity >= (1ull << 32))
        return fail(CAMA_EINVAL, "%llu band entries exceed 32-bit offsets: render fewer frames per call",
                    (unsigned long long)L.capacity);
    if ((uint64_t)F * C * L.nseg >= (1ull << 32)) return fail(CAMA_EINVAL, "too many stamp segments: render fewer frames per call");
    if ((size_t)F * C * L.NB >= (1ull << 31)) return fail(CAMA_EINVAL, "too many bands");
    if (2 * radius > L.R)
        return fail(CAMA_EINVAL, "radius %d too large for the fused path (needs 2r <= band rows = %d)", radius, L.R);
    if ((size_t)C * L.NB * 8 > 64 * 1024)
        return fail(CAMA_EINVAL, "C*bands = %d*%d exceeds the per-workgroup LDS histogram", C, L.NB);
    if (align_up((size_t)L.R * (W + 2 * radius) * 4, 16) > 160 * 1024) return fail(CAMA_EINVAL, "W=%d too wide for the LDS owner table", W);
    return CAMA_OK;
}
extern "C++" int cama_impl::check_render(int64_t N, int32_t F, int32_t C, int32_t W, int32_t H, int32_t radius, const void *scratch,
                        size_t scratch_bytes, ScratchLayout &L)
{
    if (!scratch) return fail(CAMA_EINVAL, "scratch is NULL");
    if (int rc = check_common(N, F, C, W, H)) return rc;
    if (radius < 0 || radius > CAMA_MAX_RADIUS) return fail(CAMA_EINVAL, "radius %d out of range", radius);
    const ScratchRef sc = legacy_scratch(scratch, scratch_bytes, N, F, C, H, W, radius);
    layout_scratch(N, F, C, H, W, radius, L);
    if (scratch_bytes < L.total) return fail(CAMA_EINVAL, "scratch too small: %zu < %zu", scratch_bytes, L.total);
    return check_render(N, F, C, W, H, radius, sc, L);
}

int cama_map_bounds_block(void) { return 64; }      // one box per wave of the projection kernel

int cama_map_bounds(const void *x, const void *y, const void *z, int32_t xyz_is_f64, int64_t N, double *bounds,
                    void *stream)
{
    if (N < 0) return fail(CAMA_EINVAL, "N=%lld", (long long)N);
    if (N == 0) return CAMA_OK;
    if (!x || !y || !z || !bounds) return fail(CAMA_EINVAL, "NULL pointer argument");
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK));
    if (xyz_is_f64)
        hipLaunchKernelGGL(k_block_bounds<double>, grid, dim3(BLOCK), 0, (hipStream_t)stream, x, y, z, N, bounds);
    else
        hipLaunchKernelGGL(k_block_bounds<float>, grid, dim3(BLOCK), 0, (hipStream_t)stream, x, y, z, N, bounds);
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

// Completion events attached to the NEXT overlay / scatter launch itself (hipExtLaunchKernelGGL stop event) instead of a
// separate hipEventRecord marker packet behind it: one packet less between consecutive overlays on the pipeline's
// stream.  Set by cama_pipeline_render, consumed (and cleared) by the launch.

}  // extern "C"

// cama_bin_frames / cama_bin_scenes.  scenes_dev != nullptr: multi-scene launch -- F counts ALL frames of the launch,
// frames_per_scene of them per scene, N = the largest scene's vertex count, x .. K are ignored (per-scene, from the table).
// Two halves: bin_prepass (the cull: camera masks, work lists, and what a PLANNED launch needs to size its stamp part) and
// bin_main (projection, scans, scatter).  bin_impl = both on one stream, unplanned.
// TEST HOOKS of the binning chain (tests/test_gpu_fuzz.py forces each of the production code paths on inputs that would not
// select it by themselves): CAMA_TEST_HOOKS, read once per process.  None can change a result.
struct BinEnv { bool no_cam_mask, no_candidates, no_plan; };
static const BinEnv &bin_env()
{
    static const BinEnv e{test_hook("no_cam_mask", nullptr), test_hook("no_candidates", nullptr), test_hook("no_plan", nullptr)};
    return e;
}
bool cama_impl::bin_no_plan() { return bin_env().no_plan; }
void cama_impl::bin_plan_from_demand(BinPlan &plan, const ScratchLayout &L, uint32_t most_blocks_per_frame, uint64_t chains)
{
    plan.planned = true;
    plan.nseg = std::max(most_blocks_per_frame, 1u) * (BLOCK / SEG);
    plan.capacity = std::max<uint64_t>(chains, 1) * SEG * (uint64_t)L.bands_per_stamp;
}

// does this launch's cull go through work lists (site-sized maps) / the candidate pre-pass (what a plan needs)?
extern "C++" bool cama_impl::bin_uses_list(const BinCall &b)
{
    const unsigned vblocks = (unsigned)((b.N + BLOCK - 1) / BLOCK);
    return b.N && b.block_bounds && !bin_env().no_cam_mask && (b.flags & CAMA_BIN_WORKLIST) &&
           (uint64_t)vblocks * (uint64_t)b.F >= cull_list_threshold();
}
extern "C++" bool cama_impl::bin_plannable(const BinCall &b) { return !b.scenes_dev && bin_uses_list(b) && !bin_env().no_candidates; }

extern "C++" int cama_impl::check_bin_call(const BinCall &b)
{
    if (!b.w2c || !b.crop || (!b.scenes_dev && (!b.c2cam || !b.K))) return fail(CAMA_EINVAL, "NULL pointer argument");
    if (!b.scenes_dev && b.N && (!b.x || !b.y || !b.z || (!b.colour_id && !b.draw_key))) return fail(CAMA_EINVAL, "NULL vertex buffer");
    return CAMA_OK;
}

// The cull pre-pass into the plan part `pbase` (layout L: only its plan offsets are used, so L may still be unplanned).
extern "C++" int cama_impl::bin_prepass(const BinCall &b, const ScratchLayout &L, char *pbase, hipStream_t s)
{
    if (b.F == 0) return CAMA_OK;
    const int64_t N = b.N;
    const int F = b.F, C = b.C;
    uint32_t *work_count = (uint32_t *)(pbase + L.work_count), *work = (uint32_t *)(pbase + L.work);
    // (maps without a block index have no pre-pass: nothing reads the plan part, nothing to clear -- a memset per launch is
    // 6-8 us of host time, which the 960x540 pipeline is bound by)
    if (!(N && b.block_bounds && !bin_env().no_cam_mask)) return CAMA_OK;
    // the work-list / candidate counters, the per-frame item counters and the demand word are adjacent: one memset
    HIP_TRY(hipMemsetAsync(pbase + L.work_count, 0, L.plan_zero_bytes - L.work_count, s));
    const unsigned vblocks = (unsigned)((N + BLOCK - 1) / BLOCK);
    const bool use_list = bin_uses_list(b);
    const dim3 lgrid(persistent_workgroups());
    Crop cr;
    memcpy(cr.v, b.crop, sizeof(cr.v));
    uint16_t *cam_mask = (uint16_t *)(pbase + L.cam_mask);
    const uint32_t nsub = (uint32_t)((N + 63) / 64);
    const dim3 cgrid((4 * vblocks + BLOCK - 1) / BLOCK, (unsigned)F);
    double *cam_fn = (double *)(pbase + L.cam_fn);
    // site-sized maps: a six-comparison world-space test first, the exact tests on the compacted candidates only
    const bool use_cand = use_list && !bin_env().no_candidates;
    double *frame_box = (double *)(pbase + L.frame_box);
    const unsigned fn_threads = CAMA_MAX_CAMERAS * 20 <= 256 ? 256 : 512;
    hipLaunchKernelGGL(k_camera_functionals, dim3(use_cand ? 1 + ((unsigned)F + fn_threads - 1) / fn_threads : 1),
                       dim3(fn_threads), 0, s, b.c2cam, b.K, C, b.W, b.H, cam_fn, b.w2c, (uint32_t)F, cr, frame_box);
    if (use_cand) {
        uint32_t *cand_count = work_count + CAND_COUNT_WORD, *cand = (uint32_t *)(pbase + L.cand);
        const dim3 kgrid(cgrid.x, ((unsigned)F + CAND_FRAMES - 1) / CAND_FRAMES);
        hipLaunchKernelGGL(k_block_candidates, kgrid, dim3(BLOCK), 0, s, b.block_bounds, frame_box, (uint32_t)F, vblocks, nsub,
                           (uint32_t)L.list_cap, cand_count, cand);
        hipLaunchKernelGGL(k_candidate_cameras, lgrid, dim3(BLOCK), 0, s, b.block_bounds, b.w2c, cam_fn, C, cr, vblocks, nsub,
                           cand_count, cand, cam_mask, (uint32_t)L.list_cap, work_count, work,
                           (uint32_t *)(pbase + L.frame_items), (uint32_t *)(pbase + L.work_rank),
                           (unsigned long long *)(pbase + L.demand));
    } else if (use_list)
        hipLaunchKernelGGL(k_block_cameras<true>, cgrid, dim3(BLOCK), 0, s, b.block_bounds, b.w2c, cam_fn, C, cr, vblocks, nsub,
                           cam_mask, (uint32_t)L.list_cap, work_count, work);
    else
        hipLaunchKernelGGL(k_block_cameras<false>, cgrid, dim3(BLOCK), 0, s, b.block_bounds, b.w2c, cam_fn, C, cr, vblocks, nsub,
                           cam_mask, (uint32_t)L.list_cap, work_count, work);
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

// projection -> scans (phase A) -> scatter (phase B), out of the plan part into the stamp part (layout L, planned or not).
// Segment records: the caller reads the scans' grand total between A and B and sizes the sorted buffer from it.
extern "C++" int cama_impl::bin_main(const BinCall &b, const ScratchLayout &L, const ScratchRef &sc, hipStream_t s, int phases)
{
    if (b.F == 0) return CAMA_OK;
    char *pbase = sc.plan, *sbase = sc.stamp;
    const int64_t N = b.N;
    const int F = b.F, C = b.C, W = b.W, H = b.H, radius = b.radius;
    uint32_t *counts = (uint32_t *)(sbase + L.counts), *cursor = (uint32_t *)(sbase + L.cursor);
    uint32_t *bin_off = (uint32_t *)(sbase + L.bin_off), *fc_total = (uint32_t *)(sbase + L.fc_total);
    uint32_t *fc_base = (uint32_t *)(sbase + L.fc_base);
    const int nfc = F * C;

    // counts, cursor and the segment count table are adjacent: one memset
    if (phases & BIN_PHASE_A) HIP_TRY(hipMemsetAsync(counts, 0, L.zero_bytes, s));

    FrameArgs a{};
    a.scenes = reinterpret_cast<const SceneRef *>(b.scenes_dev); a.frames_per_scene = b.frames_per_scene;
    a.x = b.x; a.y = b.y; a.z = b.z; a.colour = b.colour_id; a.key = b.draw_key; a.bounds = b.block_bounds; a.N = N;
    a.w2c = b.w2c; a.c2cam = b.c2cam; a.K = b.K; a.C = C; a.W = W; a.H = H;
    memcpy(a.crop.v, b.crop, sizeof(a.crop.v));
    a.band_shift = log2i(L.R); a.NB = L.NB; a.radius = radius;
    a.nseg = L.nseg; a.seg_cnt = (uint8_t *)(sbase + L.seg_cnt); a.stamps0 = (uint2 *)(sbase + L.stamps0);
    a.counts = counts; a.cursor = cursor; a.bin_off = bin_off; a.fc_base = fc_base;
    a.stamps = (uint2 *)sc.stamps_base(L);
    a.segments = L.segments ? (L.wu ? 2 : 1) : 0;
    if (L.wu && N >= ((int64_t)1 << 22)) return fail(CAMA_EINVAL, "anti-aliased segments: N must be below 2^22");
    if (L.segments && b.draw_key)
        return fail(CAMA_EINVAL, "segments need the map in draw order (no draw_key: a spatially sorted copy has no polyline neighbours)");
    if (L.segments && b.scenes_dev) return fail(CAMA_EINVAL, "segments are not offered for multi-scene launches");
    // XCD-aware mapping: workgroup (x, y) has linear id x + y * gridDim.x and runs on XCD id % 8.  With gridDim.x a
    // multiple of 8, vertex chunk x is processed on the SAME XCD for every frame y, so on big maps each XCD's 4 MB L2
    // keeps its 1/8 of the vertex buffer across all frames instead of re-fetching it per frame (padding workgroups
    // exit at once: no vertex, no survivor).
    const unsigned vblocks = (unsigned)((N + BLOCK - 1) / BLOCK);
    // ... and a workgroup runs several consecutive vertex blocks of its frame once there are enough workgroups to fill the
    // chip anyway (launch + histogram clear / flush per 256 vertices made big maps dispatch-bound)
    const int vb_per_wg = project_blocks_per_workgroup((uint64_t)vblocks * (uint64_t)F);
    const unsigned vchunks = (vblocks + vb_per_wg - 1) / vb_per_wg;
    const dim3 fgrid((vchunks + 7u) & ~7u, (unsigned)F);
    const size_t hist_lds = align_up((size_t)C * L.NB * 4, 16);
    // With the map's spatial index (block AABBs) a one-thread-per-(block, frame) pre-pass decided which cameras can see
    // each block at all (bin_prepass); the projection skips the others.  Site-sized maps (CAMA_BIN_WORKLIST: most
    // blocks are outside the crop box on any frame) additionally go through work lists + persistent workgroups: an empty
    // workgroup still costs ~0.8 ns of dispatch, 1.25 M of them = 1 ms.
    const bool use_list = bin_uses_list(b);
    if (L.planned && !bin_plannable(b)) return fail(CAMA_EINVAL, "a planned layout needs the candidate pre-pass");
    uint32_t *work_count = (uint32_t *)(pbase + L.work_count), *work = (uint32_t *)(pbase + L.work);
    const uint32_t *work_rank = L.planned ? (const uint32_t *)(pbase + L.work_rank) : nullptr;
    const dim3 lgrid(persistent_workgroups());
    if (N && b.block_bounds && !bin_env().no_cam_mask) {
        a.cam_mask = (const uint64_t *)(pbase + L.cam_mask);
        a.vblocks = vblocks;
    }
    if (phases & BIN_PHASE_A) {
    // live timing (cama_profile_enable): the projection takes an event pair as its own start / stop events
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (N && g_prof.on) {
        pe0 = prof_event();
        pe1 = prof_event();
        if (!pe0 || !pe1) pe0 = pe1 = nullptr;
    }
    if (N && use_list && a.cam_mask) {
        if (b.xyz_is_f64)
            hipExtLaunchKernelGGL(k_frames_project_list<double>, lgrid, dim3(BLOCK), (uint32_t)hist_lds, s, pe0, pe1, 0u, a,
                                  work_count, work, vblocks, (uint32_t)L.list_cap, work_rank);
        else
            hipExtLaunchKernelGGL(k_frames_project_list<float>, lgrid, dim3(BLOCK), (uint32_t)hist_lds, s, pe0, pe1, 0u, a,
                                  work_count, work, vblocks, (uint32_t)L.list_cap, work_rank);
        HIP_TRY(hipGetLastError());
    } else if (N) {
        if (b.xyz_is_f64)
            hipExtLaunchKernelGGL(k_frames_project<double>, fgrid, dim3(BLOCK), (uint32_t)hist_lds, s, pe0, pe1, 0u, a, vb_per_wg);
        else
            hipExtLaunchKernelGGL(k_frames_project<float>, fgrid, dim3(BLOCK), (uint32_t)hist_lds, s, pe0, pe1, 0u, a, vb_per_wg);
        HIP_TRY(hipGetLastError());
    }
    if (pe0 && pe1) g_prof.pending_project.emplace_back(pe0, pe1);
    hipLaunchKernelGGL(k_scan_bands, dim3(nfc), dim3(64), 0, s, counts, bin_off, fc_total, L.NB);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(64), 0, s, fc_total, fc_base, nfc);
    HIP_TRY(hipGetLastError());
    }
    if (N && (phases & BIN_PHASE_B)) {
        const dim3 sgrid((L.nseg + SCATTER_SEGS - 1) / SCATTER_SEGS, (unsigned)nfc);
        if (L.segments) {
            hipEvent_t stop = launch_mods().scatter_stop_event;
            launch_mods().scatter_stop_event = nullptr;
            hipExtLaunchKernelGGL(k_stamps_scatter_seg, sgrid, dim3(BLOCK), (uint32_t)align_up((size_t)2 * L.NB * 4, 16), s,
                                  nullptr, stop, 0u, a);
        } else if (launch_mods().scatter_stop_event) {
            hipExtLaunchKernelGGL(k_stamps_scatter, sgrid, dim3(BLOCK), (uint32_t)align_up((size_t)2 * L.NB * 4, 16), s,
                                  nullptr, launch_mods().scatter_stop_event, 0u, a);
            launch_mods().scatter_stop_event = nullptr;
        } else
            hipLaunchKernelGGL(k_stamps_scatter, sgrid, dim3(BLOCK), align_up((size_t)2 * L.NB * 4, 16), s, a);
        HIP_TRY(hipGetLastError());
    }
    return CAMA_OK;
}

static int bin_impl(const void *scenes_dev, int frames_per_scene, const void *x, const void *y, const void *z,
                    int32_t xyz_is_f64, const uint8_t *colour_id, const uint32_t *draw_key, const double *block_bounds,
                    int32_t flags, int64_t N, const double *w2c, int32_t F, const double *c2cam, const double *K, int32_t C,
                    const double *crop, int32_t W, int32_t H, int32_t radius, void *scratch, size_t scratch_bytes,
                    void *stream)
{
    ScratchLayout L;
    if (int rc = check_render(N, F, C, W, H, radius, scratch, scratch_bytes, L)) return rc;
    if (F == 0) return CAMA_OK;
    const BinCall b{scenes_dev, frames_per_scene, x, y, z, xyz_is_f64, colour_id, draw_key, block_bounds, flags, N, w2c, F, c2cam, K,
                    C, crop, W, H, radius};
    if (int rc = check_bin_call(b)) return rc;
    if (flags & CAMA_BIN_SEGMENTS)
        return fail(CAMA_EINVAL, "CAMA_BIN_SEGMENTS needs a pipeline that owns its scratch (cama_pipeline_render with scratch0 == NULL)");
    const ScratchRef sc = legacy_scratch(scratch, scratch_bytes, N, F, C, H, W, radius);
    if (int rc = bin_prepass(b, L, sc.plan, (hipStream_t)stream)) return rc;
    return bin_main(b, L, sc, (hipStream_t)stream);
}

extern "C" {

int cama_bin_frames(const void *x, const void *y, const void *z, int32_t xyz_is_f64, const uint8_t *colour_id,
                    const uint32_t *draw_key, const double *block_bounds, int32_t flags, int64_t N, const double *w2c, int32_t F,
                    const double *c2cam, const double *K, int32_t C,
                    const double *crop, int32_t W, int32_t H, int32_t radius, void *scratch, size_t scratch_bytes,
                    void *stream)
{
    return bin_impl(nullptr, 0, x, y, z, xyz_is_f64, colour_id, draw_key, block_bounds, flags, N, w2c, F, c2cam, K, C, crop, W, H,
                    radius, scratch, scratch_bytes, stream);
}

// Diagnostic read-back of what a finished cama_bin_frames left in `scratch` (blocks the host on `stream`): how much of the
// vertex buffer the projection actually read and how many stamps it produced -- the figures bench.py prices the
// projection's roofline with.
}  // extern "C"

extern "C++" int cama_impl::bin_stats_impl(const ScratchRef &sc, int64_t N, int32_t F, int32_t C, int32_t H, int32_t W, int32_t radius,
                          int32_t had_block_bounds, uint64_t *out, hipStream_t s)
{
    ScratchLayout L;
    if (int rc = check_render(N, F, C, W, H, radius, sc, L)) return rc;
    if (!out) return fail(CAMA_EINVAL, "out is NULL");
    out[0] = out[1] = out[2] = out[3] = 0;
    if (F == 0) return CAMA_OK;
    HIP_TRY(hipStreamSynchronize(s));
    const char *base = sc.plan, *sbase = sc.stamp;
    const size_t nfc = (size_t)F * C;
    const uint64_t vblocks = (uint64_t)((N + BLOCK - 1) / BLOCK), waves = (uint64_t)((N + 63) / 64);
    if (had_block_bounds && N && !bin_env().no_cam_mask) {
        std::vector<uint64_t> hm((size_t)vblocks * F);
        HIP_TRY(hipMemcpy(hm.data(), base + L.cam_mask, hm.size() * 8, hipMemcpyDeviceToHost));
        uint32_t wc[32];
        HIP_TRY(hipMemcpy(wc, base + L.work_count, sizeof(wc), hipMemcpyDeviceToHost));
        const auto add = [&](uint64_t m) {
            for (int w = 0; w < 4; ++w) out[0] += ((m >> (16 * w)) & 0xffffull) != 0;
            out[1] += (uint64_t)__builtin_popcountll(m);
        };
        if (wc[CAND_COUNT_WORD + 8]) {
            // candidate pre-pass: only the listed blocks' masks were written (and are read by the projection)
            std::vector<uint32_t> items(L.list_cap);
            for (int l = 0; l < 8; ++l) {
                const size_t n = wc[l] < L.list_cap ? wc[l] : L.list_cap;
                if (!n) continue;
                HIP_TRY(hipMemcpy(items.data(), base + L.work + (size_t)l * L.list_cap * 4, n * 4, hipMemcpyDeviceToHost));
                for (size_t k = 0; k < n; ++k)
                    if (items[k] < hm.size()) add(hm[items[k]]);
            }
        } else
            for (uint64_t m : hm) add(m);
    } else {
        out[0] = waves * (uint64_t)F;
        out[1] = waves * (uint64_t)F * (uint64_t)C;
    }
    std::vector<uint32_t> tot(nfc);
    HIP_TRY(hipMemcpy(tot.data(), sbase + L.fc_total, nfc * 4, hipMemcpyDeviceToHost));
    for (uint32_t v : tot) out[3] += v;                 // band entries = what the overlay reads
    // stamps = non-empty entries of the compacted segments
    std::vector<uint8_t> seg(nfc * L.nseg);
    HIP_TRY(hipMemcpy(seg.data(), sbase + L.seg_cnt, seg.size(), hipMemcpyDeviceToHost));
    for (uint8_t v : seg) out[2] += v;
    return CAMA_OK;
}

extern "C" {

int cama_bin_stats(const void *scratch, size_t scratch_bytes, int64_t N, int32_t F, int32_t C, int32_t H, int32_t W,
                   int32_t radius, int32_t had_block_bounds, uint64_t *out, void *stream)
{
    ScratchLayout L;
    if (int rc = check_render(N, F, C, W, H, radius, scratch, scratch_bytes, L)) return rc;
    return bin_stats_impl(legacy_scratch(scratch, scratch_bytes, N, F, C, H, W, radius), N, F, C, H, W, radius, had_block_bounds,
                          out, (hipStream_t)stream);
}

static thread_local uint32_t g_next_alpha256 = 256u;
extern "C++" int cama_impl::overlay_impl(const uint8_t *src, const RawSource *raw, uint8_t *mosaic, int64_t N, int32_t F, int32_t C,
                        int32_t H, int32_t W, int32_t cols, int32_t radius, const int32_t *halfwidth,
                        const uint8_t *palette_bgr, const ScratchRef &sc, void *stream,
                        const cama_scene *scenes_host, int32_t frames_per_scene)
{
    ScratchLayout L;
    if (int rc = check_render(N, F, C, W, H, radius, sc, L)) return rc;
    if (F == 0) return CAMA_OK;
    if (cols < 1) return fail(CAMA_EINVAL, "cols=%d", cols);
    const bool scenes_dev = scenes_host != nullptr;         // multi-scene launch (image pointers travel in the kernel arguments)
    if ((!scenes_dev && (!src || !mosaic)) || !palette_bgr) return fail(CAMA_EINVAL, "NULL pointer argument");
    Disc disc;
    if (make_disc(radius, halfwidth, disc)) return fail(CAMA_EINVAL, "bad radius/halfwidth table");
    // The binning chain of the NEXT launch runs beside every overlay; an overlay that fills the CU's LDS (6 workgroups of
    // 25.6 KB at W = 1600) leaves that chain one workgroup per CU.  So the overlay asks for a little more LDS than it needs --
    // just enough that ONE workgroup fewer fits a CU.  Alternating A/B runs on one box, 3 each, frames/s (sustained):
    // site map of 4e6 vertices 74.0 (76.1) -> 80.3 (84.4) k, site 1e6 106.6 (110.3) -> 107.1 (111.0) k, headline 108.3 (111.7)
    // -> 109.9 (113.2) k; 73-scene sweep, 960x540, random 1e6 within noise.  It pays where the chain is long (hundreds of
    // microseconds) and costs nothing elsewhere.
    // Round 5: "fits" has two limits -- LDS and the CU's 32 wave slots (8 workgroups of 4 waves).  At W = 960 ten workgroups
    // fit the LDS, the rule above made it nine, and the wave slots held the overlay at eight per CU = EVERY slot: the next
    // launch's binning chain got its waves only as overlay workgroups retired, its 10 us projection kernel ran 104-108 us
    // and the chain (145 us per step) -- not the 118 us overlay -- set the pace (profiles/r05_960x540_timeline.txt section 6).
    // So: fewer workgroups than what BOTH limits allow -- by `leave`, which the pipeline sets per launch from what runs beside
    // the overlay (launch_mods().overlay_leave; same-box A/Bs of 1 / 2 / 3, whole step / 8 TB/s): a stamp-heavy overlay beside a culled
    // chain (site maps, work lists) wants its occupancy: 0.760 / 0.746 / 0.733 -> 1; a long chain beside a light overlay
    // (dense 10^6-lane map) wants the room: 0.468 / 0.478 / 0.492 -> 3; everything between: 10^5 vertices 0.799 / 0.821 /
    // 0.810, headline 0.822 / 0.829, 960x540 0.728 / 0.732 / 0.739, stress 0.798 / 0.798 / 0.780 -> 2.  Launches outside a
    // pipeline (nothing runs beside them) keep 1.
    const size_t leave = (size_t)std::max(1, std::min(3, launch_mods().overlay_leave));
    launch_mods().overlay_leave = 1;
    size_t lds_pad = 0;
    {
        const size_t lds0 = align_up((size_t)L.R * (W + 2 * radius) * 4, 16), cu_lds = 160 * 1024;
        const size_t by_lds = cu_lds / lds0, by_waves = 32 / (OVERLAY_BLOCK / 64);
        const size_t fit = std::min(by_lds, by_waves);                        // workgroups per CU without padding
        if (by_lds >= 4 && by_lds <= 16 && fit > leave) {
            const size_t want = fit - leave;                                   // ... and with it
            lds_pad = align_up(cu_lds / (want + 1) + 16, 16);                 // smallest size of which `want` + 1 do not fit
            lds_pad = lds_pad > lds0 ? lds_pad - lds0 : 0;
        }
    }
    // k_overlay's owner table carries `radius` spare cells on either side of every row (rasterise_one_padded: 4-bit half
    // widths of 8 rows in one register => radius <= 7 on this path; the generic cama_stamp_points has no such limit)
    if (radius > 7) return fail(CAMA_EINVAL, "radius %d: the fused overlay supports radius <= 7", radius);
    const size_t lds = align_up((size_t)L.R * (W + 2 * radius) * 4, 16) + lds_pad;
    hipStream_t s = (hipStream_t)stream;
    const char *base = sc.stamp;              // the overlay reads the stamp part only

    OverlayArgs o{};
    o.f0 = 0;
    o.nb_magic = (uint32_t)(((1ull << 32) + (uint32_t)L.NB - 1) / (uint32_t)L.NB);
    o.cr_magic = (uint32_t)(((1ull << 32) + (uint32_t)((C + cols - 1) / cols) - 1) / (uint32_t)((C + cols - 1) / cols));
    o.cols_magic = (uint32_t)(((1ull << 32) + (uint32_t)cols - 1) / (uint32_t)cols);
    o.src = src; o.mosaic = mosaic; o.C = C; o.H = H; o.W = W; o.cols = cols; o.R = L.R; o.NB = L.NB;
    const int rows = (C + cols - 1) / cols;
    o.mosaic_row_bytes = (size_t)cols * W * 3;
    o.mosaic_frame_bytes = (size_t)rows * H * o.mosaic_row_bytes;
    o.counts = (const uint32_t *)(base + L.counts); o.bin_off = (const uint32_t *)(base + L.bin_off);
    o.fc_base = (const uint32_t *)(base + L.fc_base); o.stamps = (const uint2 *)sc.stamps_base(L);
    o.disc = disc; o.pal = make_palette(palette_bgr, g_next_alpha256);
    g_next_alpha256 = 256u;
    // (multi-scene launches: the caller has checked every scene's src / mosaic alignment and W % 16 == 0)
    const bool vec = (W % 16 == 0) && (scenes_dev || ((((raw ? 0 : (uintptr_t)src)) | (uintptr_t)mosaic) % 16 == 0));
    if (vec) {
        o.cpr = (uint32_t)(W * 3 / 16);
        o.cpr_magic = (uint32_t)(((1ull << 32) + o.cpr - 1) / o.cpr);
    }
    if (raw) {
        if (!vec) return fail(CAMA_EINVAL, "the raw-frame overlay needs W %% 16 == 0 and a 16-byte aligned mosaic");
        if (raw->H0 < 1 || raw->W0 < 1 || (int64_t)raw->H0 * raw->W0 * 3 < 8 || !raw->mapx || !raw->mapy)
            return fail(CAMA_EINVAL, "bad raw-frame source");
        o.H0 = raw->H0; o.W0 = raw->W0; o.mapx = raw->mapx; o.mapy = raw->mapy;
        o.mapx_cam = raw->separable ? W : (int64_t)H * W;
        o.mapy_cam = raw->separable ? H : (int64_t)H * W;
        o.ms.xr = raw->separable ? 0 : W; o.ms.xc = 1; o.ms.yr = raw->separable ? 1 : W; o.ms.yc = raw->separable ? 0 : 1;
        o.ms.w_magic = 0;
    }
    // items in the order (frame, camera row, band, camera column); the kernels map workgroups to items XCD-contiguously
    // (overlay_kernels.hpp: decode_band), grid = items rounded up to a multiple of 8
    const size_t items_per_frame = (size_t)rows * cols * L.NB;
    const unsigned nblocks = (unsigned)((size_t)F * items_per_frame);
    // bytes one launch touches: its frames (raw or pre-resized) + its mosaic
    const size_t frames_in_launch = scenes_dev ? (size_t)frames_per_scene : (size_t)F;
    const size_t launch_bytes = frames_in_launch * (size_t)C * 3 *
                                ((raw ? (size_t)raw->H0 * raw->W0 : (size_t)H * W) + (size_t)H * W);
    o.chunk_log2 = overlay_chunk_log2();
    // big launches of the plain overlay: the process's own choice between the contiguous and the chunked order, or a trial of
    // one of them (MapTuner); a launch that is being profiled, or that cannot carry events of its own, just follows
    const bool plain_vec = !raw && o.pal.alpha256 == 256u && vec;
    MapTrial trial{o.chunk_log2, nullptr, nullptr, 0, 0};
    const bool probe = g_overlay_probe;
    g_overlay_probe = false;
    if (probe) {
        if (!plain_vec || scenes_dev) return fail(CAMA_EINVAL, "cama_overlay_probe needs W %% 16 == 0 and 16-byte aligned buffers");
        o.chunk_log2 = MAP_CONTIGUOUS;
    } else if (launch_bytes >= MAP_BIG_LAUNCH && overlay_forced_chunk_log2() < 0 && plain_vec) {
        trial = g_map_tuner.pick(scenes_dev ? (const void *)scenes_host[0].src : (const void *)src,
                                 scenes_dev ? (const void *)scenes_host[0].mosaic : (const void *)mosaic, launch_bytes,
                                 !g_prof.on);
        o.chunk_log2 = trial.chunk_log2;
    }
    const uint32_t chunk_log2 = o.chunk_log2;
    const auto grid8 = [chunk_log2](size_t items) { return overlay_grid(items, chunk_log2); };
    o.items = nblocks;
    const dim3 ogrid = grid8(nblocks);
    if (lds > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_overlay<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void *)k_overlay<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void *)k_overlay<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // Live timing of this launch (cama_profile_enable): the dominant kernel takes the two events as ITS OWN start / stop
    // events (hipExtLaunchKernelGGL), i.e. the kernel's duration itself, the figure rocprofv3 --kernel-trace reports;
    // the other variants are bracketed by recorded events.  A profiled launch leaves the pipeline's completion event to a
    // separate record (an event can ride on a launch only once).
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const bool exact_timing = g_prof.on && !raw && o.pal.alpha256 == 256u && vec;
    if (scenes_dev && (raw || !vec || o.pal.alpha256 != 256u))
        return fail(CAMA_EINVAL, "multi-scene launches support the plain overlay only (W %% 16 == 0, opaque, pre-resized frames)");
    if (g_prof.on) {
        ev0 = prof_event();
        ev1 = prof_event();
        if (ev0 && ev1 && !exact_timing) HIP_TRY(hipEventRecord(ev0, s));
    }
    // LDS-staged raw variant when the maps are separable, the host supplied the per-band source rows, rows are
    // 16-byte multiples and the staging buffer fits; else the gather variant
    size_t lds_raw = 0;
    bool raw_lds = false;
    int Wt = 0;
    if (raw && raw->separable && raw->band_rows && raw->tile_bytes && raw->max_src_rows > 0 && raw->tiles_x >= 1 &&
        W % raw->tiles_x == 0 && (W / raw->tiles_x) % 16 == 0 && raw->max_tile_bytes % 16 == 0 &&
        ((size_t)raw->W0 * 3) % 16 == 0 && ((uintptr_t)src % 16 == 0)) {
        Wt = W / raw->tiles_x;
        lds_raw = (((size_t)L.R * Wt + 3) & ~(size_t)3) * 4 + (((size_t)Wt + 3) & ~(size_t)3) * 4 +
                  (size_t)raw->max_src_rows * raw->max_tile_bytes + 16;
        raw_lds = lds_raw <= 160 * 1024;
    }
    if (L.segments) {
        // EXTENSION: 16-byte records, discs + one-pixel segments; the plain vectorised overlay only
        if (raw || !vec || o.pal.alpha256 != 256u || scenes_dev)
            return fail(CAMA_EINVAL, "segments: plain overlay only (W %% 16 == 0, opaque, pre-resized frames, one scene)");
        if (lds > 64 * 1024) {
            HIP_TRY(hipFuncSetAttribute((const void *)k_overlay<true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_TRY(hipFuncSetAttribute((const void *)k_overlay<true, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (trial.e0) { g_map_tuner.abandon(trial); trial.e0 = trial.e1 = nullptr; }
        hipEvent_t e0 = (exact_timing && ev0 && ev1) ? ev0 : nullptr;
        hipEvent_t e1 = e0 ? ev1 : launch_mods().overlay_stop_event;
        if (L.wu)
            hipExtLaunchKernelGGL((k_overlay<true, false, false, true, true>), ogrid, dim3(OVERLAY_BLOCK), (uint32_t)lds, s, e0, e1, 0u, o);
        else
            hipExtLaunchKernelGGL((k_overlay<true, false, false, true>), ogrid, dim3(OVERLAY_BLOCK), (uint32_t)lds, s, e0, e1, 0u, o);
        if (!e0) launch_mods().overlay_stop_event = nullptr;
    } else if (raw_lds) {
        if (lds_raw > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)k_overlay_rawlds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds_raw));
        o.items = nblocks * (unsigned)raw->tiles_x;
        hipLaunchKernelGGL(k_overlay_rawlds, grid8(o.items), dim3(RAWLDS_BLOCK), lds_raw, s, o,
                           reinterpret_cast<const int2 *>(raw->band_rows),
                           reinterpret_cast<const int2 *>(raw->tile_bytes), raw->tiles_x, Wt,
                           (uint32_t)(((1ull << 32) + (uint32_t)raw->tiles_x - 1) / (uint32_t)raw->tiles_x));
    } else if (raw)
        hipLaunchKernelGGL((k_overlay<true, true>), ogrid, dim3(OVERLAY_BLOCK), lds, s, o);
    else if (o.pal.alpha256 != 256u) {      // translucent extension: its own instantiations, the exact kernels stay lean
        if (vec)
            hipLaunchKernelGGL((k_overlay<true, false, true>), ogrid, dim3(OVERLAY_BLOCK), lds, s, o);
        else
            hipLaunchKernelGGL((k_overlay<false, false, true>), ogrid, dim3(OVERLAY_BLOCK), lds, s, o);
    } else if (scenes_dev) {
        // One overlay launch PER SCENE out of the shared scratch (frames f0 .. f0 + frames_per_scene of the chain).  A single
        // launch over all scenes was built first (grid.z over every frame, per-scene image pointers in the kernel arguments)
        // and measured SLOWER: the kernel's bandwidth falls with the length of the launch -- one scene of 20 / 40 / 80 /
        // 160 / 640 frames per launch: 0.80 / 0.76 / 0.74 / 0.61-0.71 / 0.67 of 8 TB/s; 73 scenes as one launch 0.70 against
        // 0.77 as 73 launches -- so the ~12 us kernel boundary every 40 frames is the cheaper price.
        const int S = F / frames_per_scene;
        o.items = (uint32_t)((size_t)frames_per_scene * items_per_frame);
        const dim3 sgrid = grid8(o.items);
        for (int k = 0; k < S; ++k) {
            o.src = scenes_host[k].src;
            o.mosaic = scenes_host[k].mosaic;
            o.f0 = k * frames_per_scene;
            hipEvent_t e0 = (k == 0 && exact_timing && ev0 && ev1) ? ev0 : nullptr;
            hipEvent_t e1 = (k == S - 1) ? ((exact_timing && ev0 && ev1) ? ev1 : launch_mods().overlay_stop_event) : nullptr;
            if (k == 0 && trial.e0 && S > 1) {           // a mapping trial: the first scene's launch, timed on its own
                e0 = trial.e0;
                e1 = trial.e1;
            }
            hipExtLaunchKernelGGL((k_overlay<true, false>), sgrid, dim3(OVERLAY_BLOCK), (uint32_t)lds, s, e0, e1, 0u, o);
        }
        if (trial.e0 && S > 1) g_map_tuner.submitted(trial, (double)launch_bytes);
        else if (trial.e0) g_map_tuner.abandon(trial);
        if (!(exact_timing && ev0 && ev1)) launch_mods().overlay_stop_event = nullptr;
    } else if (vec && probe) {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)k_overlay_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_overlay_probe, ogrid, dim3(OVERLAY_BLOCK), lds, s, o);
    } else if (vec) {
        if (trial.e0) {
            // a mapping trial: the launch carries the tuner's start / stop events; the pipeline's completion event, which
            // would have ridden on the launch, is recorded behind it
            hipExtLaunchKernelGGL((k_overlay<true, false>), ogrid, dim3(OVERLAY_BLOCK), (uint32_t)lds, s, trial.e0, trial.e1, 0u, o);
            g_map_tuner.submitted(trial, (double)launch_bytes);
            if (launch_mods().overlay_stop_event) {
                HIP_TRY(hipEventRecord(launch_mods().overlay_stop_event, s));
                launch_mods().overlay_stop_event = nullptr;
            }
        } else if (exact_timing && ev0 && ev1) {
            hipExtLaunchKernelGGL((k_overlay<true, false>), ogrid, dim3(OVERLAY_BLOCK), (uint32_t)lds, s, ev0, ev1, 0u, o);
        } else if (launch_mods().overlay_stop_event) {
            hipExtLaunchKernelGGL((k_overlay<true, false>), ogrid, dim3(OVERLAY_BLOCK), (uint32_t)lds, s, nullptr,
                                  launch_mods().overlay_stop_event, 0u, o);
            launch_mods().overlay_stop_event = nullptr;
        } else
            hipLaunchKernelGGL((k_overlay<true, false>), ogrid, dim3(OVERLAY_BLOCK), lds, s, o);
    } else
        hipLaunchKernelGGL((k_overlay<false, false>), ogrid, dim3(OVERLAY_BLOCK), lds, s, o);
    HIP_TRY(hipGetLastError());
    if (ev0 && ev1) {
        if (!exact_timing) HIP_TRY(hipEventRecord(ev1, s));
        g_prof.pending.emplace_back(ev0, ev1);
    }
    return CAMA_OK;
}

int cama_overlay_frames(const uint8_t *src, uint8_t *mosaic, int64_t N, int32_t F, int32_t C, int32_t H, int32_t W,
                        int32_t cols, int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                        const void *scratch, size_t scratch_bytes, void *stream)
{
    return overlay_impl(src, nullptr, mosaic, N, F, C, H, W, cols, radius, halfwidth, palette_bgr,
                        legacy_scratch(scratch, scratch_bytes, N, F, C, H, W, radius), stream);
}

int cama_overlay_frames_alpha(const uint8_t *src, uint8_t *mosaic, int64_t N, int32_t F, int32_t C, int32_t H, int32_t W,
                              int32_t cols, int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                              int32_t alpha256, const void *scratch, size_t scratch_bytes, void *stream)
{
    if (alpha256 < 0 || alpha256 > 256) return fail(CAMA_EINVAL, "alpha256=%d out of range [0, 256]", alpha256);
    g_next_alpha256 = (uint32_t)alpha256;
    const int rc = overlay_impl(src, nullptr, mosaic, N, F, C, H, W, cols, radius, halfwidth, palette_bgr,
                                legacy_scratch(scratch, scratch_bytes, N, F, C, H, W, radius), stream);
    g_next_alpha256 = 256u;
    return rc;
}

// Mean duration of `reps` stamp-free overlay launches src -> mosaic (XCD-contiguous order), after one untimed launch; blocks.
int cama_overlay_probe(const uint8_t *src, uint8_t *mosaic, int32_t F, int32_t C, int32_t H, int32_t W, int32_t cols,
                       int32_t reps, double *ms_mean, void *stream)
{
    if (!src || !mosaic || !ms_mean) return fail(CAMA_EINVAL, "NULL pointer argument");
    if (F < 1 || reps < 1 || reps > 1000) return fail(CAMA_EINVAL, "F=%d reps=%d", F, reps);
    if (int rc = check_common(0, F, C, W, H)) return rc;
    constexpr int radius = 2;
    hipStream_t s = (hipStream_t)stream;
    const size_t sb = cama_render_scratch_bytes(0, F, C, H, W, radius);
    const size_t mats = align_up((size_t)F * 128 + (size_t)C * 200, 256);
    char *buf = nullptr;
    HIP_TRY(hipMalloc((void **)&buf, sb + mats));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = CAMA_OK;
    const auto done = [&](int r) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipFree(buf);
        return r;
    };
    if (hipMemsetAsync(buf + sb, 0, mats, s) != hipSuccess) return done(fail(CAMA_EHIP, "hipMemsetAsync failed"));
    const double *w2c = (const double *)(buf + sb), *c2cam = w2c + (size_t)F * 16, *K = c2cam + (size_t)C * 16;
    const double crop[6] = {-1, 1, -1, 1, -1, 1};
    rc = bin_impl(nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, w2c, F, c2cam, K, C, crop, W, H, radius,
                  buf, sb, stream);                                  // an empty map: every band's stamp count is zero
    if (rc) return done(rc);
    int32_t hw[CAMA_MAX_RADIUS + 1];
    cama_circle_halfwidths(radius, hw);
    const uint8_t pal[6] = {0, 0, 0, 0, 0, 0};
    const ScratchRef sc = legacy_scratch(buf, sb, 0, F, C, H, W, radius);
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return done(fail(CAMA_EHIP, "hipEventCreate failed"));
    for (int k = 0; k <= reps && !rc; ++k) {
        if (k == 1 && hipEventRecord(e0, s) != hipSuccess) rc = fail(CAMA_EHIP, "hipEventRecord failed");
        g_overlay_probe = true;
        if (!rc) rc = overlay_impl(src, nullptr, mosaic, 0, F, C, H, W, cols, radius, hw, pal, sc, stream);
        g_overlay_probe = false;
    }
    if (rc) { (void)hipStreamSynchronize(s); return done(rc); }
    float ms = 0.f;
    if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
        hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
        return done(fail(CAMA_EHIP, "timing the probe launches failed: %s", hipGetErrorString(hipGetLastError())));
    *ms_mean = (double)ms / reps;
    return done(CAMA_OK);
}

int cama_overlay_frames_raw(const uint8_t *raw, int32_t H0, int32_t W0, const float *mapx, const float *mapy,
                            int32_t separable, const int32_t *band_src_rows, int32_t max_src_rows,
                            const int32_t *tile_src_bytes, int32_t tiles_x, int32_t max_tile_bytes, uint8_t *mosaic,
                            int64_t N, int32_t F, int32_t C, int32_t H, int32_t W, int32_t cols, int32_t radius,
                            const int32_t *halfwidth, const uint8_t *palette_bgr, const void *scratch,
                            size_t scratch_bytes, void *stream)
{
    const RawSource rs{H0, W0, mapx, mapy, separable, band_src_rows, max_src_rows, tile_src_bytes, tiles_x, max_tile_bytes};
    return overlay_impl(raw, &rs, mosaic, N, F, C, H, W, cols, radius, halfwidth, palette_bgr,
                        legacy_scratch(scratch, scratch_bytes, N, F, C, H, W, radius), stream);
}

// cvRound(v * 32) as cv2.remap's fixed-point path does on float32 maps (round half to even)
static inline long raw35_q(float v) { return lrintf(v * 32.0f); }

int cama_raw35_plan(const float *mapx, const float *mapy, int32_t C, int32_t H, int32_t W, int32_t H0, int32_t W0,
                    uint32_t *vrows, int32_t *band_rows, int32_t *max_src_rows)
{
    if (!mapx || !mapy || !vrows || !band_rows || !max_src_rows || C < 1 || H < 1 || W < 1 || H0 < 1 || W0 < 1 || H0 > 65535)
        return fail(CAMA_EINVAL, "bad arguments");
    *max_src_rows = 0;
    // 12-pixel units, 16-byte chunk rows on both sides, the last unit's 20 source pixels exist, one unit per thread
    const int R = band_rows_for(W), NB = (H + R - 1) / R;
    if (W % 48 != 0 || ((int64_t)W0 * 3) % 16 != 0 || 5l * (W / 3) > W0 || R * (W / 12) > RAW35_MAX_BLOCK) return 0;
    static const int off[3] = {0, 1, 3}, frac[3] = {0, 21, 11};
    int most = 0;
    for (int c = 0; c < C; ++c) {
        for (int x = 0; x < W; ++x) {
            const long sx = raw35_q(mapx[(size_t)c * W + x]);
            const long x0 = sx >> 5, a = sx & 31;
            if (x0 != 5l * (x / 3) + off[x % 3] || a != frac[x % 3]) return 0;
            if (x0 < 0 || x0 + (a ? 1 : 0) >= W0) return 0;       // every weighted tap inside the frame
        }
        long prev = -1;
        for (int y = 0; y < H; ++y) {
            const long sy = raw35_q(mapy[(size_t)c * H + y]);
            const long yy0 = sy >> 5, bw = sy & 31;
            const uint32_t wt = (yy0 >= 0 && yy0 < H0) ? 32u - (uint32_t)bw : 0u;
            const uint32_t wb = (yy0 + 1 >= 0 && yy0 + 1 < H0) ? (uint32_t)bw : 0u;
            const long r0 = yy0 < 0 ? 0 : yy0 >= H0 ? H0 - 1 : yy0;
            const long r1 = yy0 + 1 < 0 ? 0 : yy0 + 1 >= H0 ? H0 - 1 : yy0 + 1;
            if (r0 < prev) return 0;                              // rows must not go backwards (bands stage a range)
            prev = r0;
            // a zero-weight bottom tap re-reads the top row instead of touching another source row
            vrows[((size_t)c * H + y) * 2] = (uint32_t)r0 | ((uint32_t)(wb ? r1 : r0) << 16);
            vrows[((size_t)c * H + y) * 2 + 1] = wt | (wb << 8);
        }
        for (int b = 0; b < NB; ++b) {
            const int ya = b * R, yb = std::min(H, ya + R) - 1;
            const uint32_t first = vrows[((size_t)c * H + ya) * 2] & 0xffffu;
            uint32_t last = first;
            for (int y = ya; y <= yb; ++y) last = std::max(last, vrows[((size_t)c * H + y) * 2] >> 16);
            band_rows[((size_t)c * NB + b) * 2] = (int32_t)first;
            band_rows[((size_t)c * NB + b) * 2 + 1] = (int32_t)(last - first + 1);
            most = std::max(most, (int)(last - first + 1));
        }
    }
    // the output of a band (R rows of W*3 bytes) is transposed through the staging area
    if ((size_t)most * W0 < (size_t)R * W) return 0;
    // round 4: the same for half bands (R / 2 rows per workgroup, cama_overlay_frames_raw35 with option raw35_subrows): their
    // largest source-row count rides in the high half of the plan word (0 = half bands cannot be used)
    int most_half = 0;
    if (R % 2 == 0 && R >= 2) {
        const int S = R / 2;
        for (int c = 0; c < C && most_half >= 0; ++c)
            for (int y = 0; y < H; y += S) {
                const int ye = std::min(H, std::min(y + S, (y / R) * R + R)) - 1;
                const int first = (int)(vrows[((size_t)c * H + y) * 2] & 0xffffu), last = (int)(vrows[((size_t)c * H + ye) * 2] >> 16);
                most_half = std::max(most_half, last - first + 1);
            }
        if ((size_t)most_half * W0 < (size_t)S * W || most_half > 0x7fff) most_half = 0;
    }
    *max_src_rows = most | (most_half << 16);
    return 1;
}

}  // extern "C"

extern "C++" int cama_impl::raw35_impl(const uint8_t *raw, int32_t H0, int32_t W0, const uint32_t *vrows, const int32_t *band_rows,
                      int32_t max_src_rows, uint8_t *mosaic, int64_t N, int32_t F, int32_t C, int32_t H, int32_t W,
                      int32_t cols, int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                      const ScratchRef &sc, void *stream)
{
    ScratchLayout L;
    if (int rc = check_render(N, F, C, W, H, radius, sc, L)) return rc;
    if (F == 0) return CAMA_OK;
    if (cols < 1) return fail(CAMA_EINVAL, "cols=%d", cols);
    if (!raw || !vrows || !band_rows || !mosaic || !palette_bgr) return fail(CAMA_EINVAL, "NULL pointer argument");
    // the plan word: low half = source rows of the tallest band, high half = of the tallest half band (0: not available)
    const int32_t max_half_rows = (max_src_rows >> 16) & 0x7fff;
    max_src_rows &= 0xffff;
    int upr = W / 12;
    const unsigned items = (unsigned)L.R * (unsigned)upr;
    if (W % 48 != 0 || ((int64_t)W0 * 3) % 16 != 0 || (uintptr_t)raw % 16 != 0 || (uintptr_t)mosaic % 16 != 0 ||
        H0 < 1 || H0 > 65535 || 5ll * (W / 3) > W0 || items > RAW35_MAX_BLOCK || max_src_rows < 1 ||
        (size_t)max_src_rows * W0 < (size_t)L.R * W)
        return fail(CAMA_EINVAL, "the 3:5 raw overlay needs W %% 48 == 0, 5*W/3 <= W0, 16-byte aligned source rows and "
                                 "mosaic and a plan from cama_raw35_plan (W=%d, W0=%d, rows=%d)", W, W0, max_src_rows);
    Disc disc;
    if (make_disc(radius, halfwidth, disc)) return fail(CAMA_EINVAL, "bad radius/halfwidth table");
    hipStream_t s = (hipStream_t)stream;
    const char *base = sc.stamp;
    OverlayArgs o{};
    o.src = raw; o.mosaic = mosaic; o.C = C; o.H = H; o.W = W; o.cols = cols; o.R = L.R; o.NB = L.NB;
    const int rows = (C + cols - 1) / cols;
    o.mosaic_row_bytes = (size_t)cols * W * 3;
    o.mosaic_frame_bytes = (size_t)rows * H * o.mosaic_row_bytes;
    o.counts = (const uint32_t *)(base + L.counts); o.bin_off = (const uint32_t *)(base + L.bin_off);
    o.fc_base = (const uint32_t *)(base + L.fc_base); o.stamps = (const uint2 *)sc.stamps_base(L);
    if (L.segments) return fail(CAMA_EINVAL, "segments: plain overlay only");
    o.disc = disc; o.pal = make_palette(palette_bgr);
    o.H0 = H0; o.W0 = W0;
    // one column tile per band (two 480-wide tiles measured 0.66-0.70 of 8 TB/s against 0.70-0.74: half rows are short bursts;
    // half bands of R / 2 rows per workgroup measured no gain either -- both variants are gone, profiles/r04_raw35_account.txt)
    constexpr int TX = 1;
    const int upr_t = upr / TX, Wt = W / TX;
    const uint32_t subrows = (uint32_t)L.R;
    const int stage_rows = max_src_rows;
    (void)max_half_rows;
    const unsigned block = (subrows * (unsigned)upr_t + 63u) & ~63u;
    // the owner table of a stamped band goes INTO the staging area, behind the tile's output rows (raw35_kernels.hpp)
    const size_t staging_dw = (size_t)stage_rows * upr_t * 15, owner_off = (size_t)subrows * upr_t * 9, owner_dw = (size_t)subrows * Wt;
    if (staging_dw < owner_off + owner_dw)
        return fail(CAMA_EINVAL, "the plan's %d source rows leave no room for the owner table (W=%d, W0=%d)", max_src_rows, W, W0);
    const size_t lds = staging_dw * 4;
    if (lds > 160 * 1024) return fail(CAMA_EINVAL, "W=%d too wide for the 3:5 raw overlay's LDS", W);
    if (lds > 64 * 1024)
        {
        HIP_TRY(hipFuncSetAttribute((const void *)k_overlay_raw35<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void *)k_overlay_raw35<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // bands per workgroup: 2 = the second band's source loads fly during the first band's blend
    constexpr int bands_per_wg = RAW35_PAIR_DEFAULT ? 2 : 1;
    const uint32_t NBx = bands_per_wg == 2 ? (uint32_t)(L.NB + 1) / 2 : (uint32_t)L.NB;
    const uint32_t nbx_magic = (uint32_t)(((1ull << 32) + NBx - 1) / NBx);
    o.items = (uint32_t)((size_t)F * rows * cols * NBx * TX);
    // (the 3:5 raw overlay always takes the chunked order unless one is forced: measured inside single processes at 1.41 GB per
    // launch, contiguous 0.63-0.71 of 8 TB/s against 0.72 for chunks of 32 bands in every process)
    o.chunk_log2 = overlay_chunk_log2();
    const dim3 rgrid = overlay_grid(o.items, o.chunk_log2);
    o.nb_magic = (uint32_t)(((1ull << 32) + (uint32_t)L.NB - 1) / (uint32_t)L.NB);
    o.cr_magic = (uint32_t)(((1ull << 32) + (uint32_t)rows - 1) / (uint32_t)rows);
    const uint32_t tx_magic = (uint32_t)(((1ull << 32) + (uint32_t)TX - 1) / (uint32_t)TX);
    const uint32_t cpt = (uint32_t)upr_t * 15u / 4u, cpt_magic = (uint32_t)(((1ull << 32) + cpt - 1) / cpt);
    o.cols_magic = (uint32_t)(((1ull << 32) + (uint32_t)cols - 1) / (uint32_t)cols);
    const int upr_full = upr;
    (void)upr_full;
    upr = upr_t;                             // the kernel works in tile units from here on
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (g_prof.on) {                         // timed launch: the events are the kernel's own start / stop (see overlay_impl)
        ev0 = prof_event();
        ev1 = prof_event();
    }
    hipEvent_t e0 = (ev0 && ev1) ? ev0 : nullptr;
    hipEvent_t e1 = e0 ? ev1 : launch_mods().overlay_stop_event;
    const uint2 *vr2 = reinterpret_cast<const uint2 *>(vrows);
    const int2 *br2 = reinterpret_cast<const int2 *>(band_rows);
    if (bands_per_wg == 2)
        hipExtLaunchKernelGGL(k_overlay_raw35<2>, rgrid, dim3(block), (uint32_t)lds, s, e0, e1, 0u, o, vr2, br2, upr, max_src_rows,
                              (int)owner_off, TX, tx_magic, cpt_magic, nbx_magic);
    else
        hipExtLaunchKernelGGL(k_overlay_raw35<1>, rgrid, dim3(block), (uint32_t)lds, s, e0, e1, 0u, o, vr2, br2, upr, stage_rows,
                              (int)owner_off, TX, tx_magic, cpt_magic, nbx_magic);
    if (!e0) launch_mods().overlay_stop_event = nullptr;
    HIP_TRY(hipGetLastError());
    if (ev0 && ev1) g_prof.pending.emplace_back(ev0, ev1);
    return CAMA_OK;
}

extern "C" {

int cama_overlay_frames_raw35(const uint8_t *raw, int32_t H0, int32_t W0, const uint32_t *vrows, const int32_t *band_rows,
                              int32_t max_src_rows, uint8_t *mosaic, int64_t N, int32_t F, int32_t C, int32_t H, int32_t W,
                              int32_t cols, int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                              const void *scratch, size_t scratch_bytes, void *stream)
{
    ScratchLayout L;
    if (int rc = check_render(N, F, C, W, H, radius, scratch, scratch_bytes, L)) return rc;
    return raw35_impl(raw, H0, W0, vrows, band_rows, max_src_rows, mosaic, N, F, C, H, W, cols, radius, halfwidth, palette_bgr,
                      legacy_scratch(scratch, scratch_bytes, N, F, C, H, W, radius), stream);
}

int cama_render_frames(const void *x, const void *y, const void *z, int32_t xyz_is_f64, const uint8_t *colour_id,
                       const uint32_t *draw_key, const double *block_bounds, int32_t flags, int64_t N, const double *w2c, int32_t F,
                       const double *c2cam, const double *K, int32_t C,
                       const double *crop, int32_t W, int32_t H, const uint8_t *src, uint8_t *mosaic, int32_t cols,
                       int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr, void *scratch,
                       size_t scratch_bytes, void *stream)
{
    // validate the overlay half first so that nothing is enqueued when it would be rejected
    if (F > 0 && (!src || !mosaic || !palette_bgr || !halfwidth || cols < 1))
        return check_common(N, F, C, W, H) ? CAMA_EINVAL : fail(CAMA_EINVAL, "NULL pointer argument");
    if (int rc = cama_bin_frames(x, y, z, xyz_is_f64, colour_id, draw_key, block_bounds, flags, N, w2c, F, c2cam, K, C, crop, W, H, radius, scratch,
                                 scratch_bytes, stream))
        return rc;
    return cama_overlay_frames(src, mosaic, N, F, C, H, W, cols, radius, halfwidth, palette_bgr, scratch, scratch_bytes,
                               stream);
}

int cama_resample_frames(const uint8_t *src, int64_t src_stride_bytes, uint8_t *dst, int64_t dst_stride_bytes,
                         int32_t n, int32_t H0, int32_t W0, int32_t H, int32_t W, const float *mapx,
                         const float *mapy, int32_t separable, void *stream)
{
    if (n < 0 || n > 65535) return fail(CAMA_EINVAL, "n=%d out of range [0, 65535]", n);
    if (H0 < 1 || W0 < 1 || H < 1 || W < 1 || W > 65535 || (int64_t)H * W * (int64_t)W >= (1ll << 32))
        return fail(CAMA_EINVAL, "bad image sizes %dx%d -> %dx%d", W0, H0, W, H);
    if ((int64_t)H0 * W0 * 3 < 8) return fail(CAMA_EINVAL, "source frame smaller than one 8-byte tap");
    if (n == 0) return CAMA_OK;
    if (!src || !dst || !mapx || !mapy) return fail(CAMA_EINVAL, "NULL pointer argument");
    MapStride ms;
    ms.xr = separable ? 0 : W; ms.xc = 1; ms.yr = separable ? 1 : W; ms.yc = separable ? 0 : 1;
    ms.w_magic = (uint32_t)(((1ull << 32) + (uint32_t)W - 1) / (uint32_t)W);     // exact for p*W < 2^32 (checked above)
    const bool vec = (W % 16 == 0) && ((uintptr_t)dst % 16 == 0) && (dst_stride_bytes % 16 == 0);
    if (vec)
        hipLaunchKernelGGL(k_resample16,
                           dim3((unsigned)(((int64_t)H * W + BLOCK * RESAMPLE_PPT - 1) / (BLOCK * RESAMPLE_PPT)), (unsigned)n),
                           dim3(BLOCK), 0, (hipStream_t)stream, src, src_stride_bytes, dst, dst_stride_bytes, H0, W0, H,
                           W, mapx, mapy, ms);
    else
        hipLaunchKernelGGL(k_resample, dim3((unsigned)(((int64_t)H * W + BLOCK - 1) / BLOCK), (unsigned)n), dim3(BLOCK),
                           0, (hipStream_t)stream, src, src_stride_bytes, dst, dst_stride_bytes, H0, W0, H, W, mapx, mapy,
                           ms);
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

int cama_bgr_to_i420(const uint8_t *bgr, int64_t src_stride_bytes, uint8_t *i420, int64_t dst_stride_bytes, int32_t n,
                     int32_t H, int32_t W, void *stream)
{
    if (n < 0 || n > 65535) return fail(CAMA_EINVAL, "n=%d out of range [0, 65535]", n);
    if (H < 2 || W < 16 || (H & 1) || (W & 15) || (int64_t)H * W >= (1ll << 31))
        return fail(CAMA_EINVAL, "bgr->i420 needs an even H and W %% 16 == 0 (got %dx%d)", W, H);
    if (n == 0) return CAMA_OK;
    if (!bgr || !i420) return fail(CAMA_EINVAL, "NULL pointer argument");
    if ((uintptr_t)bgr % 16 || (uintptr_t)i420 % 16 || src_stride_bytes % 16 || dst_stride_bytes % 16 ||
        src_stride_bytes < (int64_t)H * W * 3 || dst_stride_bytes < (int64_t)H * W * 3 / 2)
        return fail(CAMA_EINVAL, "bgr->i420 needs 16-byte aligned buffers and strides that hold a frame");
    const int work = (W >> 4) * (H >> 1);
    hipLaunchKernelGGL(k_bgr_to_i420, dim3((unsigned)((work + BLOCK - 1) / BLOCK), (unsigned)n), dim3(BLOCK), 0,
                       (hipStream_t)stream, bgr, i420, H, W, (size_t)src_stride_bytes, (size_t)dst_stride_bytes);
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

int cama_build_static_map(const float *verts, const int32_t *seg_v0, const int32_t *seg_num, const int64_t *seg_off,
                          const uint8_t *seg_colour, int32_t S, int64_t N, int32_t lift, const void *raster,
                          int32_t raster_is_f64, int32_t rows, int32_t cols, float solution, float half_w, float half_h,
                          float cx, float cy, void *x, void *y, void *z, uint8_t *colour, void *stream)
{
    if (S < 0 || N < 0 || N >= (1ll << 30)) return fail(CAMA_EINVAL, "S=%d N=%lld out of range", S, (long long)N);
    if (N == 0) return CAMA_OK;
    if (S == 0 || !verts || !seg_v0 || !seg_num || !seg_off || !seg_colour || !x || !y || !z || !colour)
        return fail(CAMA_EINVAL, "NULL pointer argument");
    if (lift && (!raster || rows < 1 || cols < 1)) return fail(CAMA_EINVAL, "lift needs a height raster");
    MapBuildArgs a{};
    a.verts = verts; a.seg_v0 = seg_v0; a.seg_num = seg_num; a.seg_off = seg_off; a.seg_colour = seg_colour;
    a.S = S; a.N = N; a.lift = lift; a.raster = raster; a.rows = rows; a.cols = cols;
    a.solution = solution; a.half_w = half_w; a.half_h = half_h; a.cx = cx; a.cy = cy;
    a.x = x; a.y = y; a.z = z; a.colour = colour;
    const dim3 grid((unsigned)((N + BLOCK - 1) / BLOCK));
    if (lift && raster_is_f64)
        hipLaunchKernelGGL(k_build_map<double>, grid, dim3(BLOCK), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(k_build_map<float>, grid, dim3(BLOCK), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// many scenes per launch
// ------------------------------------------------------------------------------------------
// validate the host copy of the scene table; returns the largest vertex count through *nmax
extern "C++" int cama_impl::check_scenes(const cama_scene *sh, const cama_scene *sd, int32_t S, int32_t F, bool need_images, int64_t *nmax)
{
    if (!sh || !sd) return fail(CAMA_EINVAL, "scene table is NULL");
    if (S < 1 || S > 65535) return fail(CAMA_EINVAL, "S=%d out of range [1, 65535]", S);
    if (F < 0 || (int64_t)S * F > 65535) return fail(CAMA_EINVAL, "S*F = %lld frames per launch out of range [0, 65535]", (long long)S * F);
    int64_t m = 0;
    for (int k = 0; k < S; ++k) {
        const cama_scene &c = sh[k];
        if (c.N < 0 || c.N >= (1ll << 30)) return fail(CAMA_EINVAL, "scene %d: N=%lld out of range", k, (long long)c.N);
        if (c.N && (!c.x || !c.y || !c.z || (!c.colour_id && !c.draw_key))) return fail(CAMA_EINVAL, "scene %d: NULL vertex buffer", k);
        if (!c.c2cam || !c.K) return fail(CAMA_EINVAL, "scene %d: NULL calibration", k);
        if (need_images && F && (!c.src || !c.mosaic || ((uintptr_t)c.src | (uintptr_t)c.mosaic) % 16))
            return fail(CAMA_EINVAL, "scene %d: frames / mosaic are NULL or not 16-byte aligned", k);
        m = std::max(m, c.N);
    }
    *nmax = m;
    return CAMA_OK;
}

extern "C" {

int cama_bin_scenes(const cama_scene *scenes_host, const cama_scene *scenes_dev, int32_t S, int32_t xyz_is_f64,
                    const double *w2c, int32_t F, int32_t C, const double *crop, int32_t W, int32_t H, int32_t radius,
                    void *scratch, size_t scratch_bytes, void *stream)
{
    int64_t nmax = 0;
    if (int rc = check_scenes(scenes_host, scenes_dev, S, F, false, &nmax)) return rc;
    return bin_impl(scenes_dev, F, nullptr, nullptr, nullptr, xyz_is_f64, nullptr, nullptr,
                    nullptr, 0, nmax, w2c, S * F, nullptr, nullptr, C, crop, W, H, radius, scratch, scratch_bytes, stream);
}

int cama_overlay_scenes(const cama_scene *scenes_host, const cama_scene *scenes_dev, int32_t S, int32_t F, int32_t C,
                        int32_t H, int32_t W, int32_t cols, int32_t radius, const int32_t *halfwidth,
                        const uint8_t *palette_bgr, const void *scratch, size_t scratch_bytes, void *stream)
{
    int64_t nmax = 0;
    if (int rc = check_scenes(scenes_host, scenes_dev, S, F, true, &nmax)) return rc;
    if (W % 16) return fail(CAMA_EINVAL, "multi-scene launches need W %% 16 == 0 (W=%d)", W);
    if (S > CAMA_MAX_SCENES_PER_LAUNCH) return fail(CAMA_EINVAL, "S=%d: at most %d scenes per chain", S, CAMA_MAX_SCENES_PER_LAUNCH);
    return overlay_impl(nullptr, nullptr, nullptr, nmax, S * F, C, H, W, cols, radius, halfwidth, palette_bgr,
                        legacy_scratch(scratch, scratch_bytes, nmax, S * F, C, H, W, radius), stream, scenes_host, F);
}

int cama_render_scenes(const cama_scene *scenes_host, const cama_scene *scenes_dev, int32_t S, int32_t xyz_is_f64,
                       const double *w2c, int32_t F, int32_t C, const double *crop, int32_t W, int32_t H, int32_t cols,
                       int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr, void *scratch,
                       size_t scratch_bytes, void *stream)
{
    int64_t nmax = 0;
    if (int rc = check_scenes(scenes_host, scenes_dev, S, F, true, &nmax)) return rc;   // nothing enqueued when the overlay
    if (W % 16 || !palette_bgr || !halfwidth || cols < 1) return fail(CAMA_EINVAL, "bad overlay arguments"); // would be refused
    if (int rc = cama_bin_scenes(scenes_host, scenes_dev, S, xyz_is_f64, w2c, F, C, crop, W, H, radius, scratch, scratch_bytes,
                                 stream))
        return rc;
    return cama_overlay_scenes(scenes_host, scenes_dev, S, F, C, H, W, cols, radius, halfwidth, palette_bgr, scratch,
                               scratch_bytes, stream);
}

int cama_set_option(const char *name, int64_t value)
{
    if (!name) return fail(CAMA_EINVAL, "name is NULL");
    for (int k = 0; k < OPT_COUNT; ++k)
        if (!strcmp(name, g_options[k].name)) {
            (void)option(k);
            g_options[k].value.store(value, std::memory_order_relaxed);
            return CAMA_OK;
        }
    return fail(CAMA_EINVAL, "unknown option '%s'", name);
}

int cama_get_option(const char *name, int64_t *value)
{
    if (!name || !value) return fail(CAMA_EINVAL, "NULL pointer argument");
    for (int k = 0; k < OPT_COUNT; ++k)
        if (!strcmp(name, g_options[k].name)) {
            *value = option(k);
            return CAMA_OK;
        }
    return fail(CAMA_EINVAL, "unknown option '%s'", name);
}

int cama_overlay_mapping_info(int32_t *decided, int32_t *samples, double *ns_per_mb)
{
    std::lock_guard<std::mutex> lock(g_map_tuner.mu);
    g_map_tuner.poll();
    const MapTuner::Entry *e = g_map_tuner.find(g_map_tuner.last_id);       // the pair of the most recent big launch
    if (decided) *decided = overlay_forced_chunk_log2() >= 0 ? overlay_forced_chunk_log2() : (e ? e->decided : -1);
    for (int k = 0; k < 2; ++k) {
        if (samples) samples[k] = e ? e->done[k] : 0;
        if (ns_per_mb) ns_per_mb[k] = (e && e->done[k]) ? MapTuner::median(e->t[k], e->done[k]) * 1e15 : 0.0;
    }
    return CAMA_OK;
}

int cama_probe_xcd_map(uint32_t *xcd_of_block, int32_t n_blocks, void *stream)
{
    if (!xcd_of_block || n_blocks < 1 || n_blocks > (1 << 20)) return fail(CAMA_EINVAL, "bad arguments");
    hipLaunchKernelGGL(k_probe_xcd, dim3((unsigned)n_blocks), dim3(64), 0, (hipStream_t)stream, xcd_of_block);
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

size_t cama_stamp_scratch_bytes(int32_t H, int32_t W)
{
    if (H < 1 || W < 1) return 0;
    return align_up((size_t)H * W * 4, 256);
}

int cama_stamp_points(const double *vu, const uint8_t *colour_id, int64_t n, uint8_t *image, int32_t H, int32_t W,
                      int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr, void *scratch,
                      size_t scratch_bytes, void *stream)
{
    return cama_stamp_polylines(vu, colour_id, nullptr, n, image, H, W, radius, halfwidth, palette_bgr, scratch, scratch_bytes,
                                stream);
}

int cama_stamp_polylines(const double *vu, const uint8_t *colour_id, const uint8_t *link, int64_t n, uint8_t *image, int32_t H,
                         int32_t W, int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr, void *scratch,
                         size_t scratch_bytes, void *stream)
{
    if (int rc = check_common(n, 1, 1, W, H)) return rc;
    if (n == 0) return CAMA_OK;
    if (!vu || !colour_id || !image || !palette_bgr || !scratch) return fail(CAMA_EINVAL, "NULL pointer argument");
    Disc disc;
    if (make_disc(radius, halfwidth, disc)) return fail(CAMA_EINVAL, "bad radius/halfwidth table");
    const size_t need = cama_stamp_scratch_bytes(H, W);
    if (scratch_bytes < need) return fail(CAMA_EINVAL, "scratch too small: %zu < %zu", scratch_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(scratch, 0, (size_t)H * W * 4, s));
    hipLaunchKernelGGL(k_stamp_global, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, vu, colour_id,
                       n, (uint32_t *)scratch, H, W, disc);
    HIP_TRY(hipGetLastError());
    if (link) {                                             // extension: segments between linked neighbours
        hipLaunchKernelGGL(k_segments_global, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, vu, colour_id, link,
                           n, (uint32_t *)scratch, H, W);
        HIP_TRY(hipGetLastError());
    }
    const int64_t npix = (int64_t)H * W;
    hipLaunchKernelGGL(k_apply_owner, dim3((unsigned)((npix + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s,
                       (const uint32_t *)scratch, image, npix, make_palette(palette_bgr));
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

// cama_stamp_polylines with ANTI-ALIASED segments: the one-image counterpart of CAMA_BIN_SEGMENTS_WU (same definition).
int cama_stamp_polylines_wu(const double *vu, const uint8_t *colour_id, const uint8_t *link, int64_t n, uint8_t *image, int32_t H,
                            int32_t W, int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr, void *scratch,
                            size_t scratch_bytes, void *stream)
{
    if (int rc = check_common(n, 1, 1, W, H)) return rc;
    if (n == 0) return CAMA_OK;
    if (!vu || !colour_id || !image || !palette_bgr || !scratch) return fail(CAMA_EINVAL, "NULL pointer argument");
    if (n >= ((int64_t)1 << 22)) return fail(CAMA_EINVAL, "anti-aliased segments: n must be below 2^22");
    Disc disc;
    if (make_disc(radius, halfwidth, disc)) return fail(CAMA_EINVAL, "bad radius/halfwidth table");
    const size_t need = cama_stamp_scratch_bytes(H, W);
    if (scratch_bytes < need) return fail(CAMA_EINVAL, "scratch too small: %zu < %zu", scratch_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(scratch, 0, (size_t)H * W * 4, s));
    hipLaunchKernelGGL(k_stamp_global_wu, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, vu, colour_id, link, n,
                       (uint32_t *)scratch, H, W, disc);
    HIP_TRY(hipGetLastError());
    const int64_t npix = (int64_t)H * W;
    hipLaunchKernelGGL(k_apply_owner_wu, dim3((unsigned)((npix + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s,
                       (const uint32_t *)scratch, image, npix, make_palette(palette_bgr));
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

}  // extern "C"
