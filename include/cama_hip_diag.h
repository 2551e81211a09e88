/*
 * cama_hip_diag.h -- the part of libcama_hip.so's C ABI that is NOT the contract: diagnostics, live kernel timing for
 * roofline reporting, the tuning options and the test hooks.  Everything a binding of the reprojection path needs is in
 * cama_hip.h (INTEGRATION.md lists those entry points one by one); what is declared here reads state back, times kernels or
 * selects among schedules that are bijections over the same work -- none of it can change a result.  bench.py, tools/ and the
 * test-suite use it; a drop-in user does not have to.
 *
 * Test hooks: ONE environment variable, CAMA_TEST_HOOKS = "name[=value],...", read once per process by the library
 * (tests/test_gpu_fuzz.py forces each production code path on inputs that would not select it by themselves):
 *   project_vb=n      vertex blocks per projection workgroup (otherwise 1..8, from the launch's size)
 *   no_cam_mask       no per-wave camera masks (every camera's chain runs for every wave inside the crop box)
 *   no_candidates     site-sized maps: the one-kernel cull instead of the candidate pre-pass
 *   no_plan           pipeline-owned scratch sized for the worst case instead of from the cull's demand
 *   <option>=v        the start value of an option of cama_set_option (below)
 */
#ifndef CAMA_HIP_DIAG_H
#define CAMA_HIP_DIAG_H

#include "cama_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* out[8] (host): launches issued, launches that were planned, buffer (re)allocations so far, scratch bytes owned, the last
 * launch's plan: segments per (frame, camera), band-entry capacity (0, 0 when it was not planned); launches so far that ran with
 * 8-row instead of 4-row bands (chosen per launch from the map's measured stamp density), rows per band of the last launch. */
int cama_pipeline_info(cama_pipeline *p, uint64_t *out);
/* cama_bin_stats (below) of the pipeline's LAST launch out of its own scratch; blocks until that launch is over. */
int cama_pipeline_bin_stats(cama_pipeline *p, uint64_t *out /* host, 4 */);
/* Test hook: the pipeline-owned stamp buffers sit between two 1 MiB zones filled with 0x5A; *bad_bytes = how many of
 * those bytes no longer hold the pattern (0 = no launch wrote outside its demand-sized buffers).  Blocks. */
int cama_pipeline_guard_check(cama_pipeline *p, int64_t *bad_bytes /* host */);

/*
 * Diagnostic: xcd_of_block[L] (device, n_blocks uint32) = the XCD (HW_REG_XCC_ID, 0..7) that block L of a 1-D grid of
 * n_blocks 64-thread blocks ran on.  The overlay kernels' XCD-contiguous workgroup -> band mapping assumes L % 8 -- for
 * speed only, the output never depends on it -- and bench.py prints what the box does.
 */
int cama_probe_xcd_map(uint32_t *xcd_of_block, int32_t n_blocks, void *stream);

/* Which workgroup -> band order do big overlay launches (>= 1.75 GiB touched) use?  The speed of the XCD-contiguous order (31)
 * depends on the buffers a launch walks (their physical placement: 0.75 .. 0.835 of 8 TB/s at 40 frames of 1600x900, the same
 * for a given pair of buffers every time), that of round-robin chunks of 32 bands (5) does not (0.775 .. 0.79).  So the
 * library times both on the first launches over each (frames, mosaic) pair -- three timings each, the launch's own start /
 * stop events, no host synchronisation -- and keeps the faster median for that pair (cama_hip.hip: MapTuner; the 64 most
 * recently used pairs per process; speed only, the pixels never depend on it).  This call reports the pair of the most
 * recent big launch: decided: -1 while measuring, else 31 or 5 (or the value overlay_chunk_log2 forces); samples[2],
 * ns_per_mb[2]: timings taken so far and their median time per 10^6 bytes, [0] = contiguous, [1] = chunked.  Any pointer
 * may be NULL.  (No reference counterpart.) */
int cama_overlay_mapping_info(int32_t *decided, int32_t *samples, double *ns_per_mb);

/* Process-wide options: schedules and orders only -- no option can change a result (every one of them selects among orders /
 * schedules that are bijections over the same work; the parity suite runs with each forced).  Changed at run time through
 * this API; launches already enqueued keep what they were given.  No environment variables (round 6: the per-option variables
 * and the options overlay_tune, bin_priority and pipeline_depth are retired, their A/Bs settled under profiles/).
 *   name                  meaning
 *   overlay_chunk_log2    -1 = library's choice (chunks of 32 bands; launches >= 1.75 GiB: whichever of contiguous / chunks the
 *                         process's own first launches over the buffer pair time faster); 0 = workgroup L renders band L;
 *                         1..30 = round-robin chunks of 2^k bands over the 8 XCDs; 31 = one contiguous range per XCD
 *   cull_list_min         (vertex block, frame) items from which the cull of a site-sized map goes through work lists +
 *                         persistent workgroups (default 16384)
 *   pipeline_host_wait    1 = cama_pipeline_render* waits on the HOST for a launch's binning before it queues the overlay (the
 *                         call blocks ~0.1 ms; no barrier packet between consecutive overlays: +1.5 % on the headline);
 *                         0 = stream-side wait; -1 (default) = host wait for plain-overlay launches that move >= 512 MiB
 *   band_rows             rows per band of a pipeline's plain single-scene launches: 0 (default) = per launch -- 8 instead of 4
 *                         when the same map's previous launches stamped >= 0.045 band entries per destination pixel (dense
 *                         maps: the rasteriser's LDS atomics, not HBM, bound the overlay there) --, 4 | 8 = forced
 * Test hooks (the parity suite's child processes): CAMA_TEST_HOOKS = "name[=value],..." -- any option above, and no_cam_mask,
 * no_candidates, no_plan, project_vb=n for the binning chain.
 * Unknown names: CAMA_EINVAL.  (No reference counterpart: the reference has no native code.) */
int cama_set_option(const char *name, int64_t value);
int cama_get_option(const char *name, int64_t *value);

/*
 * Live timing of the dominant kernel (the overlay) for roofline reporting.  While enabled on the calling
 * thread, every cama_render_frames call records a hipEvent pair around its overlay launch, on the stream
 * the kernel is launched on.  cama_profile_collect waits for the recorded events (host-blocking), returns
 * the summed elapsed milliseconds and the number of launches since the last collect, and recycles the events.
 */
int cama_profile_enable(int32_t on);
int cama_profile_collect(double *total_ms /* host */, int32_t *launches /* host */);
/* The same, launch by launch: the durations (ms) of the timed overlay launches since the last collect, in issue order, up to
 * `capacity` of them into ms[]; *launches = how many there were (all are drained).  bench.py reports min / mean / max. */
int cama_profile_collect_each(double *ms /* host */, int32_t capacity, int32_t *launches /* host */);
/* The same for the projection kernel (k_frames_project / k_frames_project_list) of every cama_bin_frames call made
 * while profiling was enabled: the kernel's own start / stop events. */
int cama_profile_collect_project(double *total_ms /* host */, int32_t *launches /* host */);
/*
 * Diagnostic read-back of a finished cama_bin_frames (same N, F, C, H, W, radius, scratch; had_block_bounds = whether
 * block_bounds was passed).  Blocks the host until `stream` is idle, then copies the small tables back.  out (host, 4):
 *   out[0] (wave, frame) items the projection read = 64-vertex runs whose camera mask was not 0 (all of them without
 *          block_bounds): the vertex buffer bytes the kernel really fetched are 13 B (16 B with draw_key) x 64 x out[0]
 *   out[1] camera bits set over those items (fp64 chains run = out[1] x 64 lanes)
 *   out[2] stamps written (8 B each)          out[3] band entries (what the overlay reads, 8 B each)
 */
int cama_bin_stats(const void *scratch, size_t scratch_bytes, int64_t N, int32_t F, int32_t C, int32_t H, int32_t W,
                   int32_t radius, int32_t had_block_bounds, uint64_t *out /* host, 4 */, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CAMA_HIP_DIAG_H */
