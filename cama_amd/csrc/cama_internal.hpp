// cama_internal.hpp -- what cama_pipeline.hip (the pipeline runtime) needs from cama_hip.hip (the kernels and their
// single-stream launch code).  Host side only, hidden visibility: none of this is ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstddef>
#include <cstdint>

#include "cama_common.hpp"

namespace cama_impl {
#pragma GCC visibility push(hidden)

// Scratch of one launch = two parts (round 4):
//   plan part   what the cull pre-pass writes and the chain reads: work-list / candidate counters, the camera masks, the work
//               and candidate lists, the per-frame item counters and ranks.  Its size depends on (N, F, C) only.
//   stamp part  band counters, segment counts, the per-wave compacted stamps (stamps0) and the band-sorted stamps.  Sized for
//               the worst case -- every vertex visible in every camera: 24 B per (frame, camera, vertex) -- unless the launch
//               was PLANNED: for site-sized maps the pre-pass knows, before the projection runs, how many (block, frame) items
//               survive per frame and how many (wave, camera) chains they carry, which bounds both buffers exactly (BinPlan).
// Callers that bring their own scratch (cama_render_frames & co.) get both parts in one buffer, plan part first, worst-case
// sized.  A pipeline that owns its scratch (cama_pipeline_render* with scratch0 == NULL) keeps them apart and plans.
struct BinPlan {
    bool planned = false;
    uint32_t nseg = 0;              // segments per (frame, camera): 4 x the largest number of surviving blocks of any frame
    uint64_t capacity = 0;          // band entries the sorted list must hold: (wave, camera) chains x 64 x bands per stamp
    // segment EXTENSION (CAMA_BIN_SEGMENTS): 16-byte records, and a record reaches every band its segment crosses, so no
    // a-priori bound exists: the band-sorted list lives in a buffer of its own, sized from the scans' grand total
    bool segments = false;
    bool wu = false;                // ... anti-aliased (CAMA_BIN_SEGMENTS_WU)
    bool have_sorted_capacity = false;
    uint64_t sorted_capacity = 0;
    // rows per band of THIS launch (0: band_rows_for(W)).  A pipeline picks 8 instead of 4 for launches whose map stamps the
    // image densely (cama_pipeline: BandMemo); both halves of a launch and cama_pipeline_bin_stats read it from here.
    int band_rows = 0;
};
struct ScratchLayout {
    // plan part (offsets from its base)
    size_t work_count, cam_mask, cam_fn, work, work_rank, frame_box, cand, frame_items, demand, plan_zero_bytes, plan_total;
    // stamp part (offsets from its base)
    size_t counts, cursor, seg_cnt, bin_off, fc_total, fc_base, stamps0, stamps, stamp_total;
    size_t zero_bytes;              // counts .. seg_cnt: cleared by one memset before the projection pass
    size_t list_cap;
    size_t total;                   // both parts in one buffer: plan_total + stamp_total
    uint64_t capacity;
    uint32_t nseg;
    bool planned, segments, wu;
    size_t record_bytes;            // 8, or 16 with segment records
    int R, NB, bands_per_stamp;
};


// where a launch's two scratch parts live
struct ScratchRef {
    char *plan = nullptr, *stamp = nullptr;
    size_t plan_bytes = 0, stamp_bytes = 0;
    BinPlan bin_plan;
    char *sorted = nullptr;         // segment extension only: the band-sorted 16-byte records
    size_t sorted_bytes = 0;
    char *stamps_base(const ScratchLayout &L) const { return L.segments ? sorted : stamp + L.stamps; }
};

struct BinCall {
    const void *scenes_dev; int frames_per_scene;      // scenes_dev: the device SceneRef table (project_kernels.hpp) or NULL
    const void *x, *y, *z; int32_t xyz_is_f64;
    const uint8_t *colour_id; const uint32_t *draw_key; const double *block_bounds; int32_t flags;
    int64_t N; const double *w2c; int32_t F; const double *c2cam, *K; int32_t C; const double *crop;
    int32_t W, H, radius;
};


struct RawSource {
    int H0, W0;
    const float *mapx, *mapy;
    int separable;
    const int32_t *band_rows;   // [C,NB,2] or NULL
    int max_src_rows;
    const int32_t *tile_bytes;  // [C,TX,2] or NULL
    int tiles_x, max_tile_bytes;
};

enum { BIN_PHASE_A = 1, BIN_PHASE_B = 2, BIN_PHASES_AB = 3 };
enum { OPT_CHUNK_LOG2 = 0, OPT_CULL_LIST_MIN, OPT_HOST_WAIT, OPT_BAND_ROWS, OPT_COUNT };

// per-launch modifiers the pipeline sets for the launch code (thread-local: a cama_pipeline is driven by one thread at a time).
// Behind a function: an `extern thread_local` object reached from another translation unit goes through a weak, hidden
// thread-local-init symbol, which position-independent code resolves to the library's load address instead of null.
struct LaunchMods {
    hipEvent_t overlay_stop_event = nullptr;    // the overlay launch takes this as its own stop event (hipExtLaunchKernelGGL)
    hipEvent_t scatter_stop_event = nullptr;    // ... and the binning chain's last kernel this one
    bool pipeline_raw_overlay = false;          // the pipelined launch being issued uses the raw 3:5 overlay
    int overlay_leave = 1;                      // workgroups per CU the next overlay leaves to its neighbours (overlay_impl)
};
LaunchMods &launch_mods();
// the pipeline's completion events ride on the launches themselves (hipExtLaunchKernelGGL stop events)
constexpr bool ext_events() { return true; }

int64_t option(int k);
int band_rows_for(int W);
int layout_scratch(int64_t N, int F, int C, int H, int W, int radius, ScratchLayout &L, const BinPlan *plan = nullptr);
ScratchRef legacy_scratch(const void *scratch, size_t scratch_bytes, int64_t N, int F, int C, int H, int W, int radius);
// what a PLANNED launch's stamp part must hold, from the cull pre-pass's two demand figures
void bin_plan_from_demand(BinPlan &plan, const ScratchLayout &L, uint32_t most_blocks_per_frame, uint64_t chains);
int check_common(int64_t N, int F, int C, int W, int H);
int check_render(int64_t N, int32_t F, int32_t C, int32_t W, int32_t H, int32_t radius, const ScratchRef &sc, ScratchLayout &L);
int check_render(int64_t N, int32_t F, int32_t C, int32_t W, int32_t H, int32_t radius, const void *scratch, size_t scratch_bytes,
                 ScratchLayout &L);
int check_bin_call(const BinCall &b);
int check_scenes(const cama_scene *sh, const cama_scene *sd, int32_t S, int32_t F, bool need_images, int64_t *nmax);
bool bin_uses_list(const BinCall &b);
bool bin_plannable(const BinCall &b);
bool bin_no_plan();                                      // test hook: worst-case scratch instead of planning
int bin_prepass(const BinCall &b, const ScratchLayout &L, char *pbase, hipStream_t s);
int bin_main(const BinCall &b, const ScratchLayout &L, const ScratchRef &sc, hipStream_t s, int phases = BIN_PHASES_AB);
int bin_stats_impl(const ScratchRef &sc, int64_t N, int32_t F, int32_t C, int32_t H, int32_t W, int32_t radius,
                   int32_t had_block_bounds, uint64_t *out, hipStream_t s);
int overlay_impl(const uint8_t *src, const RawSource *raw, uint8_t *mosaic, int64_t N, int32_t F, int32_t C, int32_t H, int32_t W,
                 int32_t cols, int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr, const ScratchRef &sc,
                 void *stream, const cama_scene *scenes_host = nullptr, int32_t frames_per_scene = 0);
int raw35_impl(const uint8_t *raw, int32_t H0, int32_t W0, const uint32_t *vrows, const int32_t *band_rows, int32_t max_src_rows,
               uint8_t *mosaic, int64_t N, int32_t F, int32_t C, int32_t H, int32_t W, int32_t cols, int32_t radius,
               const int32_t *halfwidth, const uint8_t *palette_bgr, const ScratchRef &sc, void *stream);

#pragma GCC visibility pop
}  // namespace cama_impl
