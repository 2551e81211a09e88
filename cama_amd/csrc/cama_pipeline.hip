// cama_pipeline.hip -- the pipeline runtime of libcama_hip.so: cama_pipeline_* (include/cama_hip.h).  No kernel lives here:
// a pipeline orders the binning chain and the overlay of consecutive launches on its own streams, owns demand-sized scratch
// and picks per-launch schedules (host wait, band height); the launches themselves are cama_hip.hip's (cama_internal.hpp).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cama_internal.hpp"

using namespace cama_impl;

extern "C" {

// ------------------------------------------------------------------------------------------
// two-stream pipeline context: binning of batch k+1 overlaps the overlay of batch k
// ------------------------------------------------------------------------------------------
// Launch k (1-based) uses scratch slot (k - 1) % depth and records done[k % RING] on s_ov after its overlay.  The ring
// serves three purposes: (i) launch k's binning waits for done[(k - depth) % RING], the overlay that last read its slot;
// (ii) cama_pipeline_completed() polls it, so the caller knows which launches' inputs (poses, frames) and outputs may
// be released -- the internal streams are invisible to the caller's allocator; (iii) it bounds the run-ahead: issuing
// launch k blocks until launch k - (RING - 2) has completed.
//
// Two scratch slots.  (A third -- the chain of launch k+2 hiding under overlays k and k+1 -- was an option in round 5 and measured
// the same everywhere once the overlay left wave slots to the chain, profiles/r05_960x540_timeline.txt section 6; retired.)
struct cama_pipeline {
    static constexpr int RING = 64;
    static constexpr int MAX_DEPTH = 2;
    int depth = 2;
    // s_pre: the cull pre-pass of PLANNED launches (site-sized maps) and their pose upload -- the call waits for it on the
    // host, and on its own stream it runs beside the previous launch's projection / scatter instead of queueing behind them
    hipStream_t s_bin = nullptr, s_ov = nullptr, s_pre = nullptr;
    hipEvent_t ready = nullptr, staged = nullptr, binned[MAX_DEPTH] = {};
    bool prev_planned = false;                  // the previous launch was planned: the next poses go up on s_pre
    uint64_t poses_for = 0;                     // the launch the staged poses belong to, and the stream they went up on
    bool poses_on_pre = false;
    hipEvent_t done[RING] = {};
    uint64_t issued = 0, completed = 0;
    // staged poses (cama_pipeline_stage_poses): a pinned host ring (one slot per in-flight launch) and one device pose
    // buffer per scratch slot: no per-call pose tensor on the caller's side, the upload rides on the binning stream
    double *pose_host = nullptr, *pose_dev[MAX_DEPTH] = {};
    size_t pose_cap = 0;                        // doubles per slot
    // scratch the pipeline owns (cama_pipeline_render* with scratch0 == NULL): per slot a plan part and a stamp part, grown
    // on demand and never shrunk; the stamp part carries `guard` pattern bytes on either side (cama_pipeline_guard_check)
    char *own_plan[MAX_DEPTH] = {}, *own_stamp[MAX_DEPTH] = {}, *own_sorted[MAX_DEPTH] = {};
    size_t own_plan_bytes[MAX_DEPTH] = {}, own_stamp_bytes[MAX_DEPTH] = {}, own_sorted_bytes[MAX_DEPTH] = {};
    static constexpr size_t GUARD = (size_t)1 << 20;
    uint64_t *demand_host = nullptr;            // pinned: [0] (wave, camera) chains, [1..] surviving blocks per frame
    size_t demand_cap = 0;                      // uint64 words
    // what the last launch of each slot looked like (cama_pipeline_bin_stats)
    struct Last { ScratchRef sc; int64_t N = 0; int32_t F = 0, C = 0, H = 0, W = 0, radius = 0; bool bounds = false; } last[MAX_DEPTH];
    int last_slot = -1;
    uint64_t planned_launches = 0, grows = 0;
    // Band height per launch (round 6).  Every radius-2 stamp lies in TWO 4-row bands but in 1.25 8-row bands on average: on a
    // map that stamps the image densely (10^6 lanes inside the crop box: 630 k band entries per 6-camera frame, one stamped
    // pixel per pixel) the overlay is bound by its rasteriser's LDS atomics, and 8-row bands cut the entries by a quarter
    // (whole step 0.51 -> 0.55 of 8 TB/s); everywhere else the owner table of an 8-row band (51 KB: three workgroups per CU
    // instead of six) costs 2-4 % (profiles/r05_dense1e6_bench.json).  So the choice is made per launch from what
    // the SAME map produced last time: behind the binning chain of a launch with >= BAND_MEMO_MIN_N vertices two words (the
    // grand total of band entries) are copied to pinned memory on the binning stream; a later launch over the same vertex
    // buffer finds the total once its event has fired -- nothing ever waits for it -- and picks 8 rows when the density,
    // normalised to 4-row bands, is >= BAND_DENSE entries per destination pixel.  First launch over a map: 4.
    static constexpr int64_t BAND_MEMO_MIN_N = 200000;
    static constexpr double BAND_DENSE = 0.045;     // dense 10^6 lanes: 0.073; 4*10^6-vertex site: 0.021; stress: 0.012
    struct BandMemo { const void *x = nullptr; int64_t N = 0; int32_t W = 0, H = 0; double density = -1.0; uint64_t age = 0; };
    static constexpr int BAND_MEMOS = 16;
    BandMemo band_memo[BAND_MEMOS];
    struct BandPending { bool on = false; const void *x = nullptr; int64_t N = 0; int32_t W = 0, H = 0, R = 0; double pixels = 0; };
    BandPending band_pending[MAX_DEPTH];
    hipEvent_t band_ev[MAX_DEPTH] = {};
    uint32_t *band_host = nullptr;              // pinned: [slot][2] = fc_base[last], fc_total[last]
    uint64_t band_clock = 0, band8_launches = 0;
};

// fold the finished read-backs into the memo (never blocks)
static void band_memo_harvest(cama_pipeline *p)
{
    for (int sl = 0; sl < cama_pipeline::MAX_DEPTH; ++sl) {
        cama_pipeline::BandPending &q = p->band_pending[sl];
        if (!q.on || hipEventQuery(p->band_ev[sl]) != hipSuccess) continue;
        q.on = false;
        const double entries = (double)p->band_host[2 * sl] + (double)p->band_host[2 * sl + 1];
        // in 4-row terms: a radius-2 stamp (5 rows) reaches 2 bands of 4 rows and 1.5 of 8
        const double density = entries * (q.R >= 8 ? 2.0 / 1.5 : 1.0) / std::max(q.pixels, 1.0);
        cama_pipeline::BandMemo *slot = nullptr;
        for (auto &m : p->band_memo)
            if (m.x == q.x && m.N == q.N && m.W == q.W && m.H == q.H) { slot = &m; break; }
        if (!slot) {
            slot = &p->band_memo[0];
            for (auto &m : p->band_memo)
                if (m.age < slot->age) slot = &m;                          // least recently used (unused ones have age 0)
            slot->x = q.x; slot->N = q.N; slot->W = q.W; slot->H = q.H;
        }
        slot->density = density;
        slot->age = ++p->band_clock;
    }
}

static int band_rows_choice(cama_pipeline *p, const BinCall &call, int radius, bool eligible)
{
    const int64_t forced = option(OPT_BAND_ROWS);
    const int base = band_rows_for(call.W);
    const auto fits = [&](int R) {
        return R >= base && 2 * radius <= R && align_up((size_t)R * (call.W + 2 * radius) * 4, 16) * 2 <= 160 * 1024;
    };
    if (!eligible) return 0;
    if (forced > 0) return (forced == 4 || forced == 8 || forced == 16) && fits((int)forced) ? (int)forced : 0;
    if (call.N < cama_pipeline::BAND_MEMO_MIN_N || base != 4 || !fits(8)) return 0;
    band_memo_harvest(p);
    for (auto &m : p->band_memo)
        if (m.x == call.x && m.N == call.N && m.W == call.W && m.H == call.H && m.density >= 0.0) {
            m.age = ++p->band_clock;
            return m.density >= cama_pipeline::BAND_DENSE ? 8 : 0;
        }
    return 0;
}

int cama_pipeline_create(cama_pipeline **out)
{
    if (!out) return fail(CAMA_EINVAL, "out is NULL");
    cama_pipeline *p = new cama_pipeline();
    // device-scope release, no timing: a default event makes the recording stream do a system-scope release after
    // an overlay that wrote ~1 GB, which showed up as ~20 us between consecutive overlays
    const unsigned flags = hipEventDisableTiming | hipEventReleaseToDevice;
    p->depth = 2;
    // The binning stream is the most urgent one: its kernels are small and latency-bound, the overlay beside them fills every
    // wave slot of the chip, and whatever the chain does not finish under the overlay shows up between two overlays
    // (960x540, two slots: whole step 0.655 -> 0.680 of 8 TB/s; 10^5 vertices 0.72 -> 0.74; headline unchanged).
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // lo = least urgent (numerically greatest)
    hipError_t e = hipStreamCreateWithPriority(&p->s_bin, hipStreamNonBlocking, hi);
    // (a CU-masked overlay stream that leaves 8 / 16 / 32 compute units to the chain was tried and is gone: masked queues
    // dispatch far slower -- whole step 0.72 -> 0.44 at 960x540, profiles/r05_960x540_timeline.txt)
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->s_ov, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->s_pre, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ready, flags);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->staged, flags);
    for (int k = 0; k < cama_pipeline::MAX_DEPTH && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&p->binned[k], flags);
    for (int k = 0; k < cama_pipeline::RING && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&p->done[k], flags);
    for (int k = 0; k < cama_pipeline::MAX_DEPTH && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&p->band_ev[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void **)&p->band_host, cama_pipeline::MAX_DEPTH * 2 * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        cama_pipeline_destroy(p);
        return fail(CAMA_EHIP, "cama_pipeline_create -> %s", hipGetErrorString(e));
    }
    *out = p;
    return CAMA_OK;
}

int cama_pipeline_destroy(cama_pipeline *p)
{
    if (!p) return CAMA_OK;
    if (p->s_bin) { (void)hipStreamSynchronize(p->s_bin); (void)hipStreamDestroy(p->s_bin); }
    if (p->s_ov) { (void)hipStreamSynchronize(p->s_ov); (void)hipStreamDestroy(p->s_ov); }
    if (p->s_pre) { (void)hipStreamSynchronize(p->s_pre); (void)hipStreamDestroy(p->s_pre); }
    if (p->ready) (void)hipEventDestroy(p->ready);
    if (p->staged) (void)hipEventDestroy(p->staged);
    for (int k = 0; k < cama_pipeline::MAX_DEPTH; ++k)
        if (p->binned[k]) (void)hipEventDestroy(p->binned[k]);
    for (int k = 0; k < cama_pipeline::RING; ++k)
        if (p->done[k]) (void)hipEventDestroy(p->done[k]);
    for (int k = 0; k < cama_pipeline::MAX_DEPTH; ++k)
        if (p->band_ev[k]) (void)hipEventDestroy(p->band_ev[k]);
    if (p->band_host) (void)hipHostFree(p->band_host);
    if (p->pose_host) (void)hipHostFree(p->pose_host);
    if (p->demand_host) (void)hipHostFree(p->demand_host);
    for (int k = 0; k < cama_pipeline::MAX_DEPTH; ++k) {
        if (p->pose_dev[k]) (void)hipFree(p->pose_dev[k]);
        if (p->own_plan[k]) (void)hipFree(p->own_plan[k]);
        if (p->own_sorted[k]) (void)hipFree(p->own_sorted[k]);
        if (p->own_stamp[k]) (void)hipFree(p->own_stamp[k] - cama_pipeline::GUARD);
    }
    delete p;
    return CAMA_OK;
}

// Copy the next launch's world->chassis matrices (HOST, float32 [F,16]: the np.linalg.inv result of
// cama/dataset.py:99, promoted to double here, exactly) into the pipeline's pinned ring and enqueue their upload on the
// binning stream, into the device pose buffer of the next launch's scratch slot.  Returns that device pointer: pass it
// as `w2c` to the next cama_pipeline_render*.  In order behind the previous user of the slot (launch k - depth): on s_bin, or --
// after a planned launch -- on the pre-pass stream s_pre.
int cama_pipeline_stage_poses(cama_pipeline *p, const float *w2c_host_f32, int32_t F, const double **w2c_dev)
{
    if (!p || !w2c_dev) return fail(CAMA_EINVAL, "NULL pointer argument");
    if (F < 0 || F > 65535) return fail(CAMA_EINVAL, "F=%d out of range [0, 65535]", F);
    if (F && !w2c_host_f32) return fail(CAMA_EINVAL, "w2c is NULL");
    constexpr uint64_t RING = cama_pipeline::RING;
    const size_t need = (size_t)std::max(F, 1) * 16;
    if (need > p->pose_cap) {                   // grow (rare): drain what may still read the old buffers
        HIP_TRY(hipStreamSynchronize(p->s_bin));
        HIP_TRY(hipStreamSynchronize(p->s_ov));
        HIP_TRY(hipStreamSynchronize(p->s_pre));
        const size_t cap = std::max(need, (size_t)64 * 16);
        if (p->pose_host) (void)hipHostFree(p->pose_host);
        for (int k = 0; k < cama_pipeline::MAX_DEPTH; ++k) {
            if (p->pose_dev[k]) (void)hipFree(p->pose_dev[k]);
            p->pose_dev[k] = nullptr;
        }
        p->pose_host = nullptr;
        p->pose_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&p->pose_host, RING * cap * sizeof(double), hipHostMallocDefault));
        for (int k = 0; k < cama_pipeline::MAX_DEPTH; ++k) HIP_TRY(hipMalloc((void **)&p->pose_dev[k], cap * sizeof(double)));
        p->pose_cap = cap;
    }
    const uint64_t k = p->issued + 1;           // the launch these poses belong to
    if (k > RING - 2) {                         // its ring slot was used by launch k - RING: long over once k - 62 is
        HIP_TRY(hipEventSynchronize(p->done[(k - (RING - 2)) % RING]));
    }
    double *h = p->pose_host + (size_t)(k % RING) * p->pose_cap;
    for (size_t i = 0; i < (size_t)F * 16; ++i) h[i] = (double)w2c_host_f32[i];
    const int slot = (int)((k - 1) % (uint64_t)p->depth);
    double *d = p->pose_dev[slot];
    // after a planned launch the next one is expected to be planned too: its poses go up on the pre-pass stream, behind the
    // chain that last read this slot's pose buffer (launch k - depth); cama_pipeline_render* orders whichever stream it did
    // not go up on behind the copy
    p->poses_on_pre = p->prev_planned;
    p->poses_for = k;
    if (p->poses_on_pre && k > (uint64_t)p->depth) HIP_TRY(hipStreamWaitEvent(p->s_pre, p->binned[slot], 0));
    if (F) HIP_TRY(hipMemcpyAsync(d, h, (size_t)F * 16 * sizeof(double), hipMemcpyHostToDevice, p->poses_on_pre ? p->s_pre : p->s_bin));
    *w2c_dev = d;
    return CAMA_OK;
}

// advance `completed` over every launch whose `done` event has fired (launches complete in order: one overlay stream)
static int pipeline_poll(cama_pipeline *p)
{
    while (p->completed < p->issued) {
        const hipError_t e = hipEventQuery(p->done[(p->completed + 1) % cama_pipeline::RING]);
        if (e == hipErrorNotReady) break;
        if (e != hipSuccess) return fail(CAMA_EHIP, "hipEventQuery -> %s", hipGetErrorString(e));
        ++p->completed;
    }
    return CAMA_OK;
}

int64_t cama_pipeline_issued(cama_pipeline *p) { return p ? (int64_t)p->issued : (int64_t)fail(CAMA_EINVAL, "pipeline is NULL"); }

int64_t cama_pipeline_completed(cama_pipeline *p)
{
    if (!p) return fail(CAMA_EINVAL, "pipeline is NULL");
    if (int rc = pipeline_poll(p)) return rc;
    return (int64_t)p->completed;
}

}  // extern "C"

// grow one of the pipeline's own buffers (rare; the old one may still be read by the overlay that last used the slot)
static int pipeline_grow(cama_pipeline *p, char **buf, size_t *have, size_t need, bool guarded, uint64_t k)
{
    if (*have >= need) return CAMA_OK;
    constexpr uint64_t RING = cama_pipeline::RING;
    if (k > (uint64_t)p->depth) HIP_TRY(hipEventSynchronize(p->done[(k - p->depth) % RING]));     // the previous user of this slot is over
    HIP_TRY(hipStreamSynchronize(p->s_bin));
    const size_t g = guarded ? cama_pipeline::GUARD : 0;
    if (*buf) HIP_TRY(hipFree(*buf - g));
    *buf = nullptr;
    *have = 0;
    // a quarter of headroom: consecutive launches of a drive need about the same, a little more or less
    const size_t bytes = align_up(need + need / 4, (size_t)2 << 20);
    char *raw = nullptr;
    const hipError_t e = hipMalloc((void **)&raw, bytes + 2 * g);
    if (e != hipSuccess) {
        // out of memory is the caller's to handle (render fewer frames per launch; memory parked in the caller's own caching
        // allocator can be released and the call repeated): a code of its own, and the sticky error is cleared
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? CAMA_ENOMEM : CAMA_EHIP, "hipMalloc of %zu bytes of pipeline scratch -> %s",
                    bytes + 2 * g, hipGetErrorString(e));
    }
    if (g) {
        HIP_TRY(hipMemsetAsync(raw, 0x5A, g, p->s_bin));
        HIP_TRY(hipMemsetAsync(raw + g + bytes, 0x5A, g, p->s_bin));
    }
    *buf = raw + g;
    *have = bytes;
    ++p->grows;
    return CAMA_OK;
}

// shared body of the pipelined renders: bin on s_bin, then `overlay(scratch, stream)` on s_ov.
// scratch0 == NULL: the pipeline's own scratch.  `plan_call` (may be NULL) describes the binning half: when it can be planned
// (site-sized map + block index: bin_plannable) the cull pre-pass runs first, the host waits for its two demand figures --
// a few tens of microseconds behind the previous launch's chain, while the overlays of earlier launches keep the GPU busy --
// and the stamp part is sized from them instead of from the worst case.
template <typename Overlay>
static int pipeline_impl(cama_pipeline *p, const BinCall &call, void *scratch0, void *scratch1, size_t scratch_bytes,
                         void *input_stream, bool overlay_takes_stop_event, Overlay overlay)
{
    if (!p) return fail(CAMA_EINVAL, "pipeline is NULL");
    const int64_t N = call.N;
    const int32_t F = call.F, C = call.C, W = call.W, H = call.H, radius = call.radius;
    const bool managed = !scratch0 && !scratch1;
    if (!managed && (!scratch0 || !scratch1)) return fail(CAMA_EINVAL, "two scratch buffers are needed (or none: pipeline-owned)");
    const bool segments = (call.flags & CAMA_BIN_SEGMENTS) != 0;
    if (segments && !managed) return fail(CAMA_EINVAL, "CAMA_BIN_SEGMENTS needs pipeline-owned scratch (scratch0 == scratch1 == NULL)");
    constexpr uint64_t RING = cama_pipeline::RING;
    const uint64_t k = p->issued + 1;                     // this launch
    // (caller-supplied scratch comes as TWO buffers: such launches alternate between them whatever the context's depth is)
    const uint64_t D = managed ? (uint64_t)p->depth : 2u;
    const int slot = (int)((k - 1) % D);
    // bound the run-ahead (and keep done[k % RING], last used by launch k - RING, free): launch k - (RING - 2) must be over
    if (k > RING - 2) {
        HIP_TRY(hipEventSynchronize(p->done[(k - (RING - 2)) % RING]));
        if (int rc = pipeline_poll(p)) return rc;
    }
    if (int rc = check_common(N, F, C, W, H)) return rc;
    if (radius < 0 || radius > CAMA_MAX_RADIUS) return fail(CAMA_EINVAL, "radius %d out of range", radius);
    if (F > 0)
        if (int rc = check_bin_call(call)) return rc;
    ScratchRef sc;
    ScratchLayout L;
    bool prepass_done = false;
    const bool raw_overlay_launch = launch_mods().pipeline_raw_overlay;
    sc.bin_plan.segments = segments;
    sc.bin_plan.wu = segments && (call.flags & CAMA_BIN_SEGMENTS_WU) != 0;
    if ((call.flags & CAMA_BIN_SEGMENTS_WU) && !segments) return fail(CAMA_EINVAL, "CAMA_BIN_SEGMENTS_WU goes with CAMA_BIN_SEGMENTS");
    if (!managed) {
        // validate before anything is enqueued, so a rejected call leaves the pipeline state untouched
        void *scratch = slot ? scratch1 : scratch0;
        if (int rc = check_render(N, F, C, W, H, radius, scratch, scratch_bytes, L)) return rc;
        sc = legacy_scratch(scratch, scratch_bytes, N, F, C, H, W, radius);
    } else {
        // plain single-scene launches only: the raw 3:5 overlay's source-row tables are built for band_rows_for(W), segment
        // records reach every band they cross, multi-scene chains are clip-sized maps
        const bool band_eligible = F > 0 && !segments && !launch_mods().pipeline_raw_overlay && !call.scenes_dev && call.x;
        sc.bin_plan.band_rows = band_rows_choice(p, call, radius, band_eligible);
        layout_scratch(N, F, C, H, W, radius, L, &sc.bin_plan);
        const bool plannable = F > 0 && bin_plannable(call) && !bin_no_plan();
        if (int rc = pipeline_grow(p, &p->own_plan[slot], &p->own_plan_bytes[slot], L.plan_total, false, k)) return rc;
        sc.plan = p->own_plan[slot];
        sc.plan_bytes = p->own_plan_bytes[slot];
        if (plannable) {
            // the pre-pass only writes the plan part, which the overlay of launch k - depth never reads: it need not wait for it --
            // only for that launch's binning chain (`binned`), which read the plan part and the pose buffer of this slot.  It
            // runs on its own stream: behind the previous launch's projection + scatter on s_bin the host wait below was
            // ~0.3 ms instead of ~0.1 ms, and the binning stream's cycle -- not the overlay -- set the pace (sites3x12)
            hipStream_t sp = p->s_pre;
            if (k > D) HIP_TRY(hipStreamWaitEvent(sp, p->binned[slot], 0));
            HIP_TRY(hipEventRecord(p->ready, (hipStream_t)input_stream));
            HIP_TRY(hipStreamWaitEvent(sp, p->ready, 0));
            if (p->poses_for == k && !p->poses_on_pre) {           // the poses went up on s_bin
                HIP_TRY(hipEventRecord(p->staged, p->s_bin));
                HIP_TRY(hipStreamWaitEvent(sp, p->staged, 0));
            }
            if (int rc = bin_prepass(call, L, sc.plan, sp)) return rc;
            const size_t words = 1 + (size_t)F;
            if (words > p->demand_cap) {
                if (p->demand_host) (void)hipHostFree(p->demand_host);
                p->demand_host = nullptr;
                p->demand_cap = 0;
                HIP_TRY(hipHostMalloc((void **)&p->demand_host, std::max(words, (size_t)1024) * 8, hipHostMallocDefault));
                p->demand_cap = std::max(words, (size_t)1024);
            }
            HIP_TRY(hipMemcpyAsync(p->demand_host, sc.plan + L.demand, 8, hipMemcpyDeviceToHost, sp));
            HIP_TRY(hipMemcpyAsync(p->demand_host + 1, sc.plan + L.frame_items, (size_t)F * 4, hipMemcpyDeviceToHost, sp));
            HIP_TRY(hipStreamSynchronize(sp));               // (everything s_pre did is complete: s_bin needs no event for it)
            const uint32_t *per_frame = (const uint32_t *)(p->demand_host + 1);
            uint32_t most = 0;
            for (int f = 0; f < F; ++f) most = std::max(most, per_frame[f]);
            bin_plan_from_demand(sc.bin_plan, L, most, p->demand_host[0]);
            layout_scratch(N, F, C, H, W, radius, L, &sc.bin_plan);
            prepass_done = true;
            ++p->planned_launches;
        }
        if (L.capacity >= (1ull << 32))
            return fail(CAMA_EINVAL, "%llu band entries exceed 32-bit offsets: render fewer frames per call",
                        (unsigned long long)L.capacity);
        if (int rc = pipeline_grow(p, &p->own_stamp[slot], &p->own_stamp_bytes[slot], L.stamp_total, true, k)) return rc;
        sc.stamp = p->own_stamp[slot];
        sc.stamp_bytes = p->own_stamp_bytes[slot];
        sc.sorted = p->own_sorted[slot];
        sc.sorted_bytes = p->own_sorted_bytes[slot];
        ScratchLayout chk;
        if (int rc = check_render(N, F, C, W, H, radius, sc, chk)) return rc;
    }
    // inputs (w2c upload, frames) are complete on the caller's stream at this point
    if (!prepass_done) {
        HIP_TRY(hipEventRecord(p->ready, (hipStream_t)input_stream));
        HIP_TRY(hipStreamWaitEvent(p->s_bin, p->ready, 0));
        if (p->poses_for == k && p->poses_on_pre) {               // the poses went up on s_pre (the previous launch was planned)
            HIP_TRY(hipEventRecord(p->staged, p->s_pre));
            HIP_TRY(hipStreamWaitEvent(p->s_bin, p->staged, 0));
        }
    }
    p->prev_planned = prepass_done;
    if (k > D) HIP_TRY(hipStreamWaitEvent(p->s_bin, p->done[(k - D) % RING], 0));   // the overlay that read this slot
    if (F > 0) {
        if (!prepass_done)
            if (int rc = bin_prepass(call, L, sc.plan, p->s_bin)) return rc;
        // the chain's last kernel (the scatter, launched whenever N > 0) carries `binned` as its own stop event
        const bool ext = ext_events() && N > 0;
        int rc = CAMA_OK;
        if (segments) {
            // a record reaches every band its segment crosses: project + scan first, read the grand total of band entries,
            // size the sorted list exactly, then scatter
            rc = bin_main(call, L, sc, p->s_bin, BIN_PHASE_A);
            if (rc) return rc;
            const size_t nfc = (size_t)F * C;
            uint32_t tail[2] = {0, 0};
            HIP_TRY(hipMemcpyAsync(&tail[0], sc.stamp + L.fc_base + (nfc - 1) * 4, 4, hipMemcpyDeviceToHost, p->s_bin));
            HIP_TRY(hipMemcpyAsync(&tail[1], sc.stamp + L.fc_total + (nfc - 1) * 4, 4, hipMemcpyDeviceToHost, p->s_bin));
            HIP_TRY(hipStreamSynchronize(p->s_bin));
            sc.bin_plan.have_sorted_capacity = true;
            sc.bin_plan.sorted_capacity = (uint64_t)tail[0] + tail[1] + 1;
            if (int rg = pipeline_grow(p, &p->own_sorted[slot], &p->own_sorted_bytes[slot],
                                       (size_t)sc.bin_plan.sorted_capacity * 16 + 16, false, k)) return rg;
            sc.sorted = p->own_sorted[slot];
            sc.sorted_bytes = p->own_sorted_bytes[slot];
            layout_scratch(N, F, C, H, W, radius, L, &sc.bin_plan);
            launch_mods().scatter_stop_event = ext ? p->binned[slot] : nullptr;
            rc = bin_main(call, L, sc, p->s_bin, BIN_PHASE_B);
        } else {
            launch_mods().scatter_stop_event = ext ? p->binned[slot] : nullptr;
            rc = bin_main(call, L, sc, p->s_bin);
        }
        if (rc) {
            launch_mods().scatter_stop_event = nullptr;
            return rc;
        }
        if (!ext || launch_mods().scatter_stop_event) HIP_TRY(hipEventRecord(p->binned[slot], p->s_bin));
        launch_mods().scatter_stop_event = nullptr;
        if (managed && !segments && !raw_overlay_launch && !call.scenes_dev && call.x && N >= cama_pipeline::BAND_MEMO_MIN_N &&
            band_rows_for(W) == 4 && !p->band_pending[slot].on) {
            // behind `binned` (the overlay does not wait for it): the grand total of band entries, for the NEXT launches over this map
            const size_t nfc = (size_t)F * C;
            HIP_TRY(hipMemcpyAsync(&p->band_host[2 * slot], sc.stamp + L.fc_base + (nfc - 1) * 4, 4, hipMemcpyDeviceToHost, p->s_bin));
            HIP_TRY(hipMemcpyAsync(&p->band_host[2 * slot + 1], sc.stamp + L.fc_total + (nfc - 1) * 4, 4, hipMemcpyDeviceToHost, p->s_bin));
            HIP_TRY(hipEventRecord(p->band_ev[slot], p->s_bin));
            cama_pipeline::BandPending &q = p->band_pending[slot];
            q.on = true; q.x = call.x; q.N = N; q.W = W; q.H = H; q.R = L.R; q.pixels = (double)F * C * H * W;
        }
        if (L.R != band_rows_for(W)) ++p->band8_launches;
    } else
        HIP_TRY(hipEventRecord(p->binned[slot], p->s_bin));
    // (overlays stay on ONE stream: two overlays in flight at once interleave their streams in DRAM -- measured
    // 106.8 k vs 109.9 k frames/s)
    hipStream_t so = p->s_ov;
    // `binned` also carries `ready` (s_bin waited for it above): one barrier packet between overlays, not two
    // (pipeline_host_wait: the wait happens here instead -- hipStreamWaitEvent on a complete event queues nothing, so the
    // overlay goes into its queue directly behind the previous one)
    const int64_t host_wait = option(OPT_HOST_WAIT);
    // (not for a planned launch: its call already waited for the cull on the host, and waiting for the rest of the chain as
    // well would start the NEXT launch's cull ~0.3 ms later than the GPU could -- sites3x12: 94 k -> 89 k frames/s)
    // Round 5: from 512 MiB (was 1 GiB) for the plain overlay -- since the overlay leaves wave slots to the chain, the chain is
    // over long before the overlay it runs beside, the wait costs the host nothing it needs (960x540 x 40 frames: host 45 ->
    // 125 us per 124 us step) and the barrier packet's ~8 us go: 0.1277 -> 0.1236 ms, twice on one box.  The raw 3:5 overlay
    // stays on the stream-side wait (0.2359 -> 0.2339 / 0.2410: no clear gain).
    const bool raw_overlay = launch_mods().pipeline_raw_overlay;
    launch_mods().pipeline_raw_overlay = false;
    if (host_wait > 0 || (host_wait < 0 && !prepass_done && !segments && !raw_overlay &&
                          (size_t)F * C * H * W * 6 >= ((size_t)1 << 29)))
        HIP_TRY(hipEventSynchronize(p->binned[slot]));
    HIP_TRY(hipStreamWaitEvent(so, p->binned[slot], 0));
    launch_mods().overlay_stop_event = (overlay_takes_stop_event && ext_events()) ? p->done[k % RING] : nullptr;
    // how much of every CU the overlay leaves to the NEXT launch's binning chain (overlay_impl: lds_pad)
    launch_mods().overlay_leave = (F > 0 && bin_uses_list(call)) ? 1 : (N >= 500000 ? 3 : 2);
    if (int rc = overlay(sc, (void *)so)) {
        launch_mods().overlay_stop_event = nullptr;
        return rc;
    }
    if (!(overlay_takes_stop_event && ext_events()) || launch_mods().overlay_stop_event)
        HIP_TRY(hipEventRecord(p->done[k % RING], so));      // the launch did not take the event: record it behind
    launch_mods().overlay_stop_event = nullptr;
    p->issued = k;
    p->last[slot].sc = sc;
    p->last[slot].N = N; p->last[slot].F = F; p->last[slot].C = C; p->last[slot].H = H; p->last[slot].W = W;
    p->last[slot].radius = radius;
    p->last[slot].bounds = call.block_bounds != nullptr;
    p->last_slot = slot;
    return CAMA_OK;
}

template <typename Overlay>
static int pipeline_render_impl(cama_pipeline *p, const void *x, const void *y, const void *z, int32_t xyz_is_f64,
                                const uint8_t *colour_id, const uint32_t *draw_key, const double *block_bounds, int32_t flags, int64_t N,
                                const double *w2c, int32_t F, const double *c2cam, const double *K, int32_t C,
                                const double *crop, int32_t W, int32_t H, int32_t radius, void *scratch0, void *scratch1,
                                size_t scratch_bytes, void *input_stream, bool overlay_takes_stop_event, Overlay overlay)
{
    const BinCall b{nullptr, 0, x, y, z, xyz_is_f64, colour_id, draw_key, block_bounds, flags, N, w2c, F, c2cam, K, C, crop, W, H,
                    radius};
    return pipeline_impl(p, b, scratch0, scratch1, scratch_bytes, input_stream, overlay_takes_stop_event, overlay);
}

extern "C" {

int cama_pipeline_render_scenes(cama_pipeline *p, const cama_scene *scenes_host, const cama_scene *scenes_dev, int32_t S,
                                int32_t xyz_is_f64, const double *w2c, int32_t F, int32_t C, const double *crop, int32_t W,
                                int32_t H, int32_t cols, int32_t radius, const int32_t *halfwidth,
                                const uint8_t *palette_bgr, void *scratch0, void *scratch1, size_t scratch_bytes,
                                void *input_stream)
{
    int64_t nmax = 0;
    if (int rc = check_scenes(scenes_host, scenes_dev, S, F, true, &nmax)) return rc;
    if (W % 16 || !palette_bgr || !halfwidth || cols < 1) return fail(CAMA_EINVAL, "bad overlay arguments");
    if (!w2c || !crop) return fail(CAMA_EINVAL, "NULL pointer argument");
    const BinCall b{scenes_dev, F, nullptr, nullptr, nullptr, xyz_is_f64, nullptr, nullptr,
                    nullptr, 0, nmax, w2c, S * F, nullptr, nullptr, C, crop, W, H, radius};
    return pipeline_impl(p, b, scratch0, scratch1, scratch_bytes, input_stream, true,
                         [&](const ScratchRef &sc, void *so) {
                             return overlay_impl(nullptr, nullptr, nullptr, nmax, S * F, C, H, W, cols, radius, halfwidth,
                                                 palette_bgr, sc, so, scenes_host, F);
                         });
}

int cama_pipeline_render(cama_pipeline *p, const void *x, const void *y, const void *z, int32_t xyz_is_f64,
                         const uint8_t *colour_id, const uint32_t *draw_key, const double *block_bounds, int32_t flags, int64_t N,
                         const double *w2c, int32_t F,
                         const double *c2cam, const double *K, int32_t C, const double *crop, int32_t W, int32_t H,
                         const uint8_t *src, uint8_t *mosaic, int32_t cols, int32_t radius, const int32_t *halfwidth,
                         const uint8_t *palette_bgr, void *scratch0, void *scratch1, size_t scratch_bytes,
                         void *input_stream)
{
    if (F > 0 && (!src || !mosaic || !palette_bgr || !halfwidth || cols < 1)) return fail(CAMA_EINVAL, "NULL pointer argument");
    return pipeline_render_impl(p, x, y, z, xyz_is_f64, colour_id, draw_key, block_bounds, flags, N, w2c, F, c2cam, K, C, crop, W, H,
                                radius, scratch0, scratch1, scratch_bytes, input_stream, true,
                                [&](const ScratchRef &sc, void *so) {
                                    return overlay_impl(src, nullptr, mosaic, N, F, C, H, W, cols, radius, halfwidth,
                                                        palette_bgr, sc, so);
                                });
}

int cama_pipeline_render_raw35(cama_pipeline *p, const void *x, const void *y, const void *z, int32_t xyz_is_f64,
                               const uint8_t *colour_id, const uint32_t *draw_key, const double *block_bounds, int32_t flags, int64_t N,
                               const double *w2c, int32_t F, const double *c2cam, const double *K, int32_t C,
                               const double *crop, int32_t W, int32_t H, const uint8_t *raw, int32_t H0, int32_t W0,
                               const uint32_t *vrows, const int32_t *band_rows, int32_t max_src_rows, uint8_t *mosaic,
                               int32_t cols, int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                               void *scratch0, void *scratch1, size_t scratch_bytes, void *input_stream)
{
    if (F > 0 && (!raw || !vrows || !band_rows || !mosaic || !palette_bgr || !halfwidth || cols < 1))
        return fail(CAMA_EINVAL, "NULL pointer argument");
    launch_mods().pipeline_raw_overlay = true;
    return pipeline_render_impl(p, x, y, z, xyz_is_f64, colour_id, draw_key, block_bounds, flags, N, w2c, F, c2cam, K, C, crop, W, H,
                                radius, scratch0, scratch1, scratch_bytes, input_stream, true,
                                [&](const ScratchRef &sc, void *so) {
                                    return raw35_impl(raw, H0, W0, vrows, band_rows, max_src_rows, mosaic, N, F, C, H, W, cols,
                                                      radius, halfwidth, palette_bgr, sc, so);
                                });
}

int cama_pipeline_render_clip(cama_pipeline *p, const cama_clip *clip, const float *w2c_host_f32, int32_t F, const uint8_t *src,
                              uint8_t *mosaic, void *input_stream, int64_t *issued, int64_t *completed)
{
    if (!p || !clip) return fail(CAMA_EINVAL, "NULL pointer argument");
    if (clip->kind != 0 && clip->kind != 1) return fail(CAMA_EINVAL, "cama_clip.kind %d (0 = frames at output size, 1 = raw 3:5)", clip->kind);
    if (clip->radius < 0 || clip->radius > CAMA_MAX_RADIUS) return fail(CAMA_EINVAL, "radius %d out of range", clip->radius);
    const double *w2c = nullptr;
    if (int rc = cama_pipeline_stage_poses(p, w2c_host_f32, F, &w2c)) return rc;
    const cama_clip &c = *clip;
    int rc;
    if (c.kind == 0)
        rc = cama_pipeline_render(p, c.x, c.y, c.z, c.xyz_is_f64, c.colour_id, c.draw_key, c.block_bounds, c.flags, c.N, w2c, F,
                                  c.c2cam, c.K, c.C, c.crop, c.W, c.H, src, mosaic, c.cols, c.radius, c.halfwidth, c.palette_bgr,
                                  nullptr, nullptr, 0, input_stream);
    else
        rc = cama_pipeline_render_raw35(p, c.x, c.y, c.z, c.xyz_is_f64, c.colour_id, c.draw_key, c.block_bounds, c.flags, c.N, w2c,
                                        F, c.c2cam, c.K, c.C, c.crop, c.W, c.H, src, c.H0, c.W0, c.vrows, c.band_rows,
                                        c.max_src_rows, mosaic, c.cols, c.radius, c.halfwidth, c.palette_bgr, nullptr, nullptr, 0,
                                        input_stream);
    if (rc) return rc;
    if (issued) *issued = (int64_t)p->issued;
    if (completed) {
        if (int rp = pipeline_poll(p)) return rp;
        *completed = (int64_t)p->completed;
    }
    return CAMA_OK;
}

int64_t cama_pipeline_scratch_bytes(cama_pipeline *p)
{
    if (!p) return fail(CAMA_EINVAL, "pipeline is NULL");
    int64_t n = 0;
    for (int k = 0; k < cama_pipeline::MAX_DEPTH; ++k)
        n += (int64_t)p->own_plan_bytes[k] + (int64_t)p->own_stamp_bytes[k] + (int64_t)p->own_sorted_bytes[k];
    return n;
}

int cama_pipeline_info(cama_pipeline *p, uint64_t *out)
{
    if (!p || !out) return fail(CAMA_EINVAL, "NULL pointer argument");
    out[0] = p->issued; out[1] = p->planned_launches; out[2] = p->grows;
    out[3] = (uint64_t)cama_pipeline_scratch_bytes(p);
    const int sl = p->last_slot;
    out[4] = sl >= 0 ? p->last[sl].sc.bin_plan.nseg : 0;
    out[5] = sl >= 0 ? p->last[sl].sc.bin_plan.capacity : 0;
    out[6] = p->band8_launches;                                  // launches that ran with other than band_rows_for(W) rows per band
    out[7] = sl >= 0 ? (uint64_t)(p->last[sl].sc.bin_plan.band_rows > 0 ? p->last[sl].sc.bin_plan.band_rows : band_rows_for(p->last[sl].W)) : 0;
    return CAMA_OK;
}

int cama_pipeline_bin_stats(cama_pipeline *p, uint64_t *out)
{
    if (!p || !out) return fail(CAMA_EINVAL, "NULL pointer argument");
    if (p->last_slot < 0) return fail(CAMA_EINVAL, "no launch yet");
    HIP_TRY(hipStreamSynchronize(p->s_ov));
    const cama_pipeline::Last &l = p->last[p->last_slot];
    return bin_stats_impl(l.sc, l.N, l.F, l.C, l.H, l.W, l.radius, l.bounds ? 1 : 0, out, p->s_bin);
}

int cama_pipeline_guard_check(cama_pipeline *p, int64_t *bad_bytes)
{
    if (!p || !bad_bytes) return fail(CAMA_EINVAL, "NULL pointer argument");
    HIP_TRY(hipStreamSynchronize(p->s_bin));
    HIP_TRY(hipStreamSynchronize(p->s_ov));
    constexpr size_t G = cama_pipeline::GUARD;
    std::vector<uint8_t> h(G);
    int64_t bad = 0;
    for (int k = 0; k < cama_pipeline::MAX_DEPTH; ++k) {
        if (!p->own_stamp[k]) continue;
        for (int side = 0; side < 2; ++side) {
            const char *g = side ? p->own_stamp[k] + p->own_stamp_bytes[k] : p->own_stamp[k] - G;
            HIP_TRY(hipMemcpy(h.data(), g, G, hipMemcpyDeviceToHost));
            for (uint8_t v : h) bad += v != 0x5A;
        }
    }
    *bad_bytes = bad;
    return CAMA_OK;
}

int cama_pipeline_join(cama_pipeline *p, void *stream)
{
    if (!p) return fail(CAMA_EINVAL, "pipeline is NULL");
    // one overlay stream: the newest launch's event covers every earlier one
    if (p->issued) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, p->done[p->issued % cama_pipeline::RING], 0));
    return CAMA_OK;
}

}  // extern "C"
