/*
 * cama_hip.h -- C ABI of libcama_hip.so: the MI355X (gfx950) implementation of the
 * CAMA multi-camera reprojection hot path.
 *
 * The reference (manymuch/CAMA) is pure Python and has no FFI of its own; the
 * drop-in boundary a user sees is the Python class surface main.py calls
 * (cama/dataset.py:11 ClipManager, cama/reproject.py:20,163 MapManager /
 * CameraManager, cama/tools.py:12 VideoGenerator).  This header is the native
 * boundary UNDER that surface: each entry point names the reference lines whose
 * third-party native call sites (numpy/BLAS matmul, boolean gathers, cv2.circle,
 * np.concatenate) it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless the parameter
 *     is documented "host"; the library never allocates, frees or synchronises.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all work is
 *     enqueued asynchronously on it.
 *   - return 0 on success; <0 on error: CAMA_EINVAL (bad argument, nothing was
 *     enqueued), CAMA_EHIP (a HIP call failed), CAMA_ENOMEM (pipeline-owned
 *     scratch could not be allocated).  cama_last_error() returns a
 *     thread-local message for the most recent failure on the calling thread.
 *   - thread-safety: calls on distinct streams / buffers may run concurrently.  The library keeps three pieces of
 *     process-wide state, none of which can change a result: (i) the options (cama_set_option; atomics; no
 *     environment variables), (ii) the overlay's per-buffer-pair choice between two workgroup
 *     orders (cama_overlay_mapping_info; a table behind a mutex, entries keyed by device and buffer addresses, its
 *     pending timing events are destroyed whether or not their launch succeeded), (iii) per-THREAD profiling state
 *     (cama_profile_*) and the per-thread last-error string.  A cama_pipeline is owned by one thread at a time.
 *   - matrices are row-major doubles.  world->chassis matrices are the float32
 *     np.linalg.inv result (cama/dataset.py:99) promoted to double on the host
 *     (exact), so one matrix type crosses the boundary.
 *   - arithmetic contract: float64, k-ordered FMA chains
 *     acc = m0*x; acc = fma(m1,y,acc); acc = fma(m2,z,acc); [acc = fma(m3,1,acc)]
 *     -- bit-identical to numpy/OpenBLAS float64 matmul on blocks of >= 2 points
 *     (what the reference executes); IEEE float64 division; comparisons and the
 *     int32 truncation exactly as the reference orders them.
 */
#ifndef CAMA_HIP_H
#define CAMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAMA_ABI_VERSION 24
#define CAMA_OK      0
#define CAMA_EINVAL (-1)
#define CAMA_EHIP   (-2)
#define CAMA_ENOMEM (-3)      /* a cama_pipeline could not grow its own scratch (hipMalloc: out of memory); nothing was enqueued,
                               * the pipeline stays usable: release memory (or render fewer frames per launch) and call again */
#define CAMA_MAX_CAMERAS 16
#define CAMA_MAX_RADIUS  15
#define CAMA_BIN_WORKLIST 1   /* flags of the bin / render entries */
#define CAMA_BIN_SEGMENTS 2   /* EXTENSION (no reference semantics; BASELINE.json's north_star asks for rasterised line segments, the
                               * reference draws a disc per point: SURVEY.md D1): a point whose colour_id byte has bit 1 set is
                               * also joined to its predecessor in the vertex buffer -- when both are visible in the camera --
                               * by a one-pixel 8-connected Bresenham segment between the two truncated pixels, under the later
                               * point's draw index and colour.  Only through cama_pipeline_render with pipeline-owned scratch
                               * (a record reaches every band its segment crosses: the sorted list is sized from the scans'
                               * grand total, which costs one host wait per launch); draw_key must be NULL. */
#define CAMA_BIN_SEGMENTS_WU 4 /* with CAMA_BIN_SEGMENTS: the segments are ANTI-ALIASED (Wu's line, the north_star's "Bresenham/Wu
                               * line-raster + blend"): stepping the major axis pixel by pixel with the minor coordinate in 16.16
                               * fixed point, the two pixels either side of the exact line get 8-bit coverages 255 - f and f; a
                               * pixel shows the claim -- disc (coverage 255) or segment -- with the greatest (draw index,
                               * coverage) and is blended ONCE over the source, (colour * a + source * (256 - a) + 128) >> 8 with
                               * a = coverage + (coverage >> 7).  Definition: oracle/cama_oracle.c, oracle_render_frame_wu.
                               * N < 2^22 (the owner cells carry the coverage in their low byte). */
#define CAMA_MAX_SCENES_PER_LAUNCH 1024  /* cama_*_scenes: scenes per chain */

int cama_abi_version(void);
const char *cama_last_error(void);

/*
 * Homogeneous transform + optional inclusive crop test, F matrices over one point set.
 * Replaces MapManager.transform_3d_instance_maps (cama/reproject.py:108-116) and the mask of
 * MapManager.crop_3d_instance_maps (cama/reproject.py:118-131) as called from
 * ClipManager.yield_frame (cama/dataset.py:99-105).
 *   xyz        [N,3] AoS, float32 (xyz_is_f64 == 0) or float64
 *   T          [F,16]
 *   crop       host pointer to {xmin,xmax,ymin,ymax,zmin,zmax} or NULL (no test)
 *   out_xyz    [F,N,3] float64 (may be NULL when only the mask is wanted)
 *   crop_mask  [F,N] uint8 1 = inside (may be NULL)
 */
int cama_transform_points(const void *xyz, int32_t xyz_is_f64, int64_t N,
                          const double *T, int32_t F, const double *crop,
                          double *out_xyz, uint8_t *crop_mask, void *stream);

/*
 * Inclusive axis-aligned box test on already-transformed points: the mask of
 * MapManager.crop_3d_instance_maps (cama/reproject.py:118-131) for a flat point list.
 *   xyz [n,3] float64, crop host[6] {xmin,xmax,ymin,ymax,zmin,zmax}, mask [n] uint8
 */
int cama_crop_points(const double *xyz, int64_t n, const double *crop, uint8_t *mask, void *stream);

/*
 * Per-camera chassis->camera transform, intrinsics, z>0, divide, in-image mask.
 * Replaces ClipManager.project_all_camera (cama/dataset.py:108-117) =
 * transform_3d_instance_maps (cama/reproject.py:108-116) + CameraManager.project_to_image
 * (cama/reproject.py:187-205) for a flat point list (compaction stays with the caller).
 *   chassis_xyz [n,3] float64      c2cam [C,16]     K [C,9] (already scaled to W x H)
 *   vu          [C,n,2] float64, (v,u) order, written for every point (inf/nan where z <= 0)
 *   vis         [C,n] uint8
 */
int cama_project_points(const double *chassis_xyz, int64_t n,
                        const double *c2cam, const double *K, int32_t C,
                        int32_t W, int32_t H, double *vu, uint8_t *vis, void *stream);

/*
 * Fused chain for F frames over the static map (structure-of-arrays float32 vertex buffer):
 * world->chassis, crop, then per camera chassis->camera, K, cull.  "Test/API mode" of the
 * fused kernel: materialises coordinates.  Same reference lines as the two calls above.
 *   x,y,z   [N] each, float32 (xyz_is_f64 == 0; the reference's dtype for a float32 BEV raster,
 *           cama/reproject.py:79,103) or float64        w2c [F,16]
 *   vu      [F,C,N,2] float64 (v,u); written where the point passed the crop
 *   vis     [F,C,N] uint8; always written
 *   crop_mask [F,N] uint8 or NULL
 */
int cama_project_frames(const void *x, const void *y, const void *z, int32_t xyz_is_f64, int64_t N,
                        const double *w2c, int32_t F,
                        const double *c2cam, const double *K, int32_t C,
                        const double *crop /* host, 6 */, int32_t W, int32_t H,
                        double *vu, uint8_t *vis, uint8_t *crop_mask, void *stream);

/*
 * Fused render of F frames: the whole per-frame hot path
 *   yield_frame body (cama/dataset.py:99-105) -> project_all_camera (:108-117) ->
 *   render_maps (cama/reproject.py:246-257: int32 truncation, colour by class, filled circle
 *   per point, sequential = last writer wins) -> concate_image (cama/tools.py:22-25).
 * No coordinates are materialised.  Stamps are binned by (frame, camera, row band), each band
 * is resolved deterministically (per-pixel max draw index) and written once into the mosaic.
 *   x,y,z      [N] float32/float64 (xyz_is_f64)   colour_id [N] uint8: bit 0 = palette index (0 lane grey, 1 gold); bit 1 =
 *              "joined to the previous vertex" (same polyline), read only with CAMA_BIN_SEGMENTS; other bits ignored
 *   draw_key   NULL, or [N] uint32 = (draw index << 1) | colour for vertex buffers stored in another order than
 *              they are drawn (e.g. spatially sorted): "last writer wins" follows the draw index, not storage
 *              order; colour_id is ignored when draw_key is given
 *   block_bounds NULL, or the map's AABBs (one per cama_map_bounds_block() = 64 consecutive vertices, what one wave
 *              projects) from cama_map_bounds().  A pre-pass then decides per (box, frame) which cameras can see it at all
 *              (conservative box-vs-frustum and box-vs-crop tests): vertices outside the crop box are skipped without
 *              being read, and for the others the fp64 projection chain runs only for the cameras that may see them
 *              (dense lane maps: 1.1 of 6 per wave).  Conservative, so the output is bit-identical with and without it
 *   flags      0, or CAMA_BIN_WORKLIST (needs block_bounds): the surviving (block, frame) items go through work lists walked
 *              by persistent workgroups instead of one workgroup per item -- for site-sized maps, where ~95 % of the
 *              blocks are outside the crop box on any frame
 *   w2c        [F,16]             c2cam [C,16]   K [C,9]   crop host[6]
 *   src        [F,C,H,W,3] uint8 BGR frames (already at output size)
 *   mosaic     [F, rows*H, cols*W, 3] uint8, rows = ceil(C/cols); camera c goes to cell
 *              (c / cols, c % cols) -- tools.py:23-24 for C = 6, cols = 3 with camera_list order
 *   radius     circle radius (reference: 2); halfwidth host[radius+1]: half-width of the filled
 *              footprint on rows +-k (OpenCV midpoint circle; data, so a real-cv2 fixture can
 *              correct it) -- see cama_circle_halfwidths(); every halfwidth[k] <= radius (else CAMA_EINVAL)
 *   palette_bgr host[2*3]
 *   scratch    device scratch of at least cama_render_scratch_bytes(...) bytes
 */
size_t cama_render_scratch_bytes(int64_t N, int32_t F, int32_t C, int32_t H, int32_t W, int32_t radius);

/*
 * Spatial index of a static map for the crop step (MapManager.crop_3d_instance_maps, cama/reproject.py:118-131, which
 * the reference evaluates for every vertex of the site map on every frame): bounds[b] = {xlo,xhi,ylo,yhi,zlo,zhi} of
 * vertices [b*B, (b+1)*B), B = cama_map_bounds_block() (64); bounds is device double[ceil(N/B)*6].  Computed once per
 * map (or per spatially sorted copy), passed as `block_bounds` to the render entries.
 */
int cama_map_bounds_block(void);
int cama_map_bounds(const void *x, const void *y, const void *z, int32_t xyz_is_f64, int64_t N, double *bounds,
                    void *stream);
int cama_render_frames(const void *x, const void *y, const void *z, int32_t xyz_is_f64,
                       const uint8_t *colour_id, const uint32_t *draw_key, const double *block_bounds, int32_t flags, int64_t N,
                       const double *w2c, int32_t F,
                       const double *c2cam, const double *K, int32_t C,
                       const double *crop, int32_t W, int32_t H,
                       const uint8_t *src, uint8_t *mosaic, int32_t cols,
                       int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                       void *scratch, size_t scratch_bytes, void *stream);

/*
 * The two halves of cama_render_frames, for callers that pipeline scenes: the binning half (count -> scan ->
 * fill; touches only the vertex buffer, matrices and scratch) of scene k+1 can run on one stream while the
 * overlay half (reads src + scratch, writes mosaic) of scene k streams on another, each scene with its own
 * scratch.  cama_render_frames == cama_bin_frames followed by cama_overlay_frames on one stream.  The caller
 * orders the overlay after its own binning (stream order or an event).  Arguments as above.
 */
int cama_bin_frames(const void *x, const void *y, const void *z, int32_t xyz_is_f64,
                    const uint8_t *colour_id, const uint32_t *draw_key, const double *block_bounds, int32_t flags, int64_t N,
                    const double *w2c, int32_t F,
                    const double *c2cam, const double *K, int32_t C,
                    const double *crop, int32_t W, int32_t H, int32_t radius,
                    void *scratch, size_t scratch_bytes, void *stream);
int cama_overlay_frames(const uint8_t *src, uint8_t *mosaic, int64_t N, int32_t F, int32_t C,
                        int32_t H, int32_t W, int32_t cols,
                        int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                        const void *scratch, size_t scratch_bytes, void *stream);

/*
 * Two-stream pipeline context for callers that render many batches (scenes, or frame ranges of a long scene) back to
 * back: cama_pipeline_render enqueues cama_bin_frames on an internal stream and cama_overlay_frames on another, so the
 * binning of batch k+1 overlaps the HBM-bound overlay of batch k; scratch0 / scratch1 (each >=
 * cama_render_scratch_bytes) are used alternately.  Inputs must be complete on `input_stream` when the call is made;
 * outputs are complete after cama_pipeline_join(p, stream) in `stream`'s order.  The context owns three HIP streams (binning,
 * overlay, and one for the cull pre-pass + pose upload of planned launches, below), a ring of 64 completion events (device-scope release, no timing), its staged-pose buffers and -- see below -- optionally
 * its scratch.  One context per thread.
 *
 * Pipeline-owned, demand-sized scratch (round 4): pass scratch0 == scratch1 == NULL (scratch_bytes ignored) to any
 * cama_pipeline_render*.  cama_render_scratch_bytes() is a worst case -- every vertex visible in every camera: 24 B per
 * (frame, camera, vertex), 24 GB for 167 frames of a 10^6-vertex map -- while a site-sized map leaves ~5 % of its vertices
 * inside the crop box (cama/reproject.py:118-131) of which a camera sees a fraction (:187-205).  The pipeline then keeps two
 * buffers per slot: the plan part (camera masks, work lists: O(F * N / 256)) and the stamp part.  For launches whose cull goes
 * through the candidate pre-pass (block_bounds + CAMA_BIN_WORKLIST) it runs that pre-pass first, waits on the host for two
 * figures it leaves behind -- the surviving blocks of the busiest frame and the (wave, camera) projection chains of all
 * surviving blocks, an exact upper bound of what the projection can emit -- and sizes the stamp part from them (grow-only,
 * hipMalloc inside the call when it has to grow); other launches get the worst case.  The pre-pass runs on the context's
 * third stream, beside the previous launch's projection and scatter (queued behind them it made the wait ~0.3 ms and the
 * binning stream, not the overlay, set the pace: 94 k -> 105 k frames/s on 12 scenes over three 10^6-vertex site maps); the
 * wait is ~0.1 ms, hidden by the overlays already queued; it makes such a call synchronous with the pre-pass, not with the
 * overlays -- AND with whatever the caller had queued on `input_stream` before the call: the pre-pass reads the map, which is
 * only known to be complete in that stream's order, so it waits for an event recorded there (a caller that queues long work
 * on its input stream -- a download, a decode -- before a planned launch blocks for it; issue planned launches from a stream
 * that carries only their inputs).  An unplanned plain-overlay launch that moves >= 512 MiB waits on the host for its whole binning chain instead of queueing a
 * stream-side wait in front of its overlay (option pipeline_host_wait, below).  cama_pipeline_info / cama_pipeline_bin_stats /
 * cama_pipeline_guard_check below report on it.
 *
 * Lifetime of what a launch reads and writes: the internal streams are invisible to the caller's allocator, so every
 * buffer handed to launch k (w2c, src, mosaic, the map, the calibration) must stay allocated and unmodified until
 * cama_pipeline_completed(p) >= k (launches are numbered 1, 2, ... = cama_pipeline_issued(p) right after the call)
 * or until a cama_pipeline_join() has been ordered before its release.  cama_pipeline_completed() never blocks
 * (hipEventQuery).  At most 62 launches are in flight: issuing launch k blocks the host until launch k - 62 is over.
 * Both return < 0 on error.
 */
typedef struct cama_pipeline cama_pipeline;
int cama_pipeline_create(cama_pipeline **out);
int cama_pipeline_destroy(cama_pipeline *p);
int cama_pipeline_render(cama_pipeline *p, const void *x, const void *y, const void *z, int32_t xyz_is_f64,
                         const uint8_t *colour_id, const uint32_t *draw_key, const double *block_bounds, int32_t flags, int64_t N,
                         const double *w2c, int32_t F,
                         const double *c2cam, const double *K, int32_t C,
                         const double *crop, int32_t W, int32_t H,
                         const uint8_t *src, uint8_t *mosaic, int32_t cols,
                         int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                         void *scratch0, void *scratch1, size_t scratch_bytes, void *input_stream);
/*
 * Host poses for the next launch: copies F world->chassis matrices from HOST memory (float32 [F,16], the np.linalg.inv
 * result of cama/dataset.py:99; promoted to double exactly) into the context's pinned ring, enqueues the upload on the
 * binning stream and returns the DEVICE address to pass as `w2c` to the next cama_pipeline_render*: no device allocation
 * or caller-side upload per launch.  The address is fixed per scratch slot.  The context allocates these two small pose
 * buffers and the pinned ring itself
 * (64 x max F x 128 bytes); everything else stays caller-owned.
 */
int cama_pipeline_stage_poses(cama_pipeline *p, const float *w2c_host_f32, int32_t F, const double **w2c_dev);
/* The same pipeline with the 3:5 raw-frame overlay (cama_overlay_frames_raw35) as its overlay half. */
int cama_pipeline_render_raw35(cama_pipeline *p, const void *x, const void *y, const void *z, int32_t xyz_is_f64,
                               const uint8_t *colour_id, const uint32_t *draw_key, const double *block_bounds, int32_t flags, int64_t N,
                               const double *w2c, int32_t F, const double *c2cam, const double *K, int32_t C,
                               const double *crop, int32_t W, int32_t H, const uint8_t *raw, int32_t H0, int32_t W0,
                               const uint32_t *vrows, const int32_t *band_rows, int32_t max_src_rows, uint8_t *mosaic,
                               int32_t cols, int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                               void *scratch0, void *scratch1, size_t scratch_bytes, void *input_stream);
/*
 * One call per launch of a clip (round 5): everything about a clip that does NOT change from launch to launch -- its static
 * map, calibration, crop box, output geometry, disc, palette, and for raw sensor frames the 3:5 tap tables -- lives in a
 * `cama_clip` the caller fills once (HOST struct of device pointers and scalars; read during the call only).  A launch is then
 *     cama_pipeline_stage_poses + cama_pipeline_render (kind 0) | cama_pipeline_render_raw35 (kind 1) + the two counters
 * in ONE crossing of the boundary: the host float32 world->chassis matrices of the launch's F frames (the np.linalg.inv result,
 * cama/dataset.py:99), the frames to read and the mosaic to write.  Pipeline-owned scratch.  *issued = this launch's number,
 * *completed = launches known to be over (as cama_pipeline_issued / cama_pipeline_completed; either may be NULL).  Replaces, per
 * step of ClipManager.render_clip, four calls with ~30 marshalled arguments each -- on launches of 0.1 ms (960x540) the host, not
 * the GPU, set the pace.  Same kernels, same results.  Mirrors the per-frame body of cama/dataset.py:78-126 like the entries above.
 */
typedef struct cama_clip {
    const void *x, *y, *z;             /* static map, SoA [N] each (float32, or float64 when xyz_is_f64) */
    const uint8_t *colour_id;          /* [N] */
    const uint32_t *draw_key;          /* [N] or NULL */
    const double *block_bounds;        /* per-64-vertex AABBs (cama_map_bounds) or NULL */
    const double *c2cam, *K;           /* [C,16], [C,9] */
    const uint32_t *vrows;             /* kind 1: cama_raw35_plan's tables */
    const int32_t *band_rows;
    int64_t N;
    double crop[6];
    int32_t xyz_is_f64, flags, C, W, H, cols, radius;
    int32_t kind;                      /* 0: `src` holds frames at output size [F,C,H,W,3]; 1: raw sensor frames [F,C,H0,W0,3] */
    int32_t H0, W0, max_src_rows, reserved;
    int32_t halfwidth[CAMA_MAX_RADIUS + 1];
    uint8_t palette_bgr[8];            /* 2 x BGR, 2 bytes of padding */
} cama_clip;
int cama_pipeline_render_clip(cama_pipeline *p, const cama_clip *clip, const float *w2c_host_f32, int32_t F,
                              const uint8_t *src, uint8_t *mosaic, void *input_stream, int64_t *issued, int64_t *completed);
int cama_pipeline_join(cama_pipeline *p, void *stream);
int64_t cama_pipeline_issued(cama_pipeline *p);
int64_t cama_pipeline_completed(cama_pipeline *p);
/* Device bytes of scratch the pipeline owns right now (all slots, plan + stamp parts). */
int64_t cama_pipeline_scratch_bytes(cama_pipeline *p);

/*
 * Many scenes per chain.  main.py:32 renders scene after scene; a scene of ~1e4 vertices x 40 frames is ~0.35 ms of GPU
 * work behind 6-7 launches, so a 73-scene sweep issues ~500 launches and at the reference's default image size the HOST
 * issuing them is the bound.  These entries render S scenes that share frame count F, camera count C, image size and vertex
 * dtype through ONE binning chain (memset, projection, scans, scatter over all S*F frames: a device table of per-scene
 * pointers, frame f of the chain = frame f % F of scene f / F) followed by one overlay launch per scene out of the shared
 * scratch.  (One overlay launch for all scenes was measured slower: the overlay's bandwidth falls with the length of a
 * launch)
 *   scenes_host / scenes_dev   the same S <= CAMA_MAX_SCENES_PER_LAUNCH entries in host memory (validated here; the
 *                              overlay launches take their image pointers from this copy) and in device memory (read by
 *                              the projection through scalar loads); both stay valid until the chain has completed
 *   w2c        [S*F,16] device, scene-major          scratch >= cama_render_scratch_bytes(max N, S*F, C, H, W, radius)
 * Restrictions: no block_bounds / work lists (meant for many small maps; big site maps are launched per scene), the
 * plain pre-resized-frame overlay only.  Bit-identical to S calls of the single-scene entries.
 */
typedef struct cama_scene {
    const void *x, *y, *z;        /* [N] SoA vertex buffer (float32, or float64 when xyz_is_f64) */
    const uint8_t *colour_id;     /* [N] */
    const uint32_t *draw_key;     /* NULL or [N] */
    const double *c2cam, *K;      /* [C,16], [C,9] */
    const uint8_t *src;           /* [F,C,H,W,3] */
    uint8_t *mosaic;              /* [F, rows*H, cols*W, 3] */
    int64_t N;
} cama_scene;
int cama_bin_scenes(const cama_scene *scenes_host, const cama_scene *scenes_dev, int32_t S, int32_t xyz_is_f64,
                    const double *w2c, int32_t F, int32_t C, const double *crop, int32_t W, int32_t H, int32_t radius,
                    void *scratch, size_t scratch_bytes, void *stream);
int cama_overlay_scenes(const cama_scene *scenes_host, const cama_scene *scenes_dev, int32_t S, int32_t F, int32_t C,
                        int32_t H, int32_t W, int32_t cols, int32_t radius, const int32_t *halfwidth,
                        const uint8_t *palette_bgr, const void *scratch, size_t scratch_bytes, void *stream);
int cama_render_scenes(const cama_scene *scenes_host, const cama_scene *scenes_dev, int32_t S, int32_t xyz_is_f64,
                       const double *w2c, int32_t F, int32_t C, const double *crop, int32_t W, int32_t H, int32_t cols,
                       int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr, void *scratch,
                       size_t scratch_bytes, void *stream);
/* The two-stream pipeline (above) with a multi-scene launch as its unit; poses may come from cama_pipeline_stage_poses
 * (S*F matrices). */
int cama_pipeline_render_scenes(cama_pipeline *p, const cama_scene *scenes_host, const cama_scene *scenes_dev, int32_t S,
                                int32_t xyz_is_f64, const double *w2c, int32_t F, int32_t C, const double *crop, int32_t W,
                                int32_t H, int32_t cols, int32_t radius, const int32_t *halfwidth,
                                const uint8_t *palette_bgr, void *scratch0, void *scratch1, size_t scratch_bytes,
                                void *input_stream);

/*
 * EXTENSION (no reference semantics; the reference draws opaque discs, cama/reproject.py:253-256): overlay half with
 * translucent stamps.  A pixel covered by stamps becomes round(alpha*colour + (1-alpha)*source) ONCE, colour = the last
 * writer's; alpha256 = alpha in 1/256, 256 == cama_overlay_frames.  Checked against the oracle's own restatement.
 */
int cama_overlay_frames_alpha(const uint8_t *src, uint8_t *mosaic, int64_t N, int32_t F, int32_t C,
                              int32_t H, int32_t W, int32_t cols,
                              int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                              int32_t alpha256, const void *scratch, size_t scratch_bytes, void *stream);

/*
 * Mosaic egress: n BGR24 frames [H,W,3] -> planar YUV 4:2:0 (I420: Y plane H*W, then U and V planes (H/2)*(W/2)), so
 * that what is downloaded and piped to the encoder is 1.5 bytes per pixel and already in the encoder's pixel format.
 * Replaces the colour conversion libswscale performs behind VideoGenerator's bgr24 pipe (cama/tools.py:13-20,27-32:
 * rawvideo bgr24 in, -pix_fmt yuv420p libx264 out).  Arithmetic = libswscale's unscaled C path for BGR24 -> YUV420P
 * (rgb24toyv12_c with the BT.601 limited-range table, 15-bit fixed point; chroma from the first pixel of the first
 * line of each 2x2 block), restated in oracle/cama_oracle.py:bgr_to_i420.  PARITY UNPINNED (no ffmpeg on either box).
 *   bgr  [n] frames at bgr + k*src_stride_bytes     i420 [n] frames at i420 + k*dst_stride_bytes
 *   H even, W % 16 == 0, buffers and strides 16-byte aligned.
 */
int cama_bgr_to_i420(const uint8_t *bgr, int64_t src_stride_bytes, uint8_t *i420, int64_t dst_stride_bytes, int32_t n,
                     int32_t H, int32_t W, void *stream);

/*
 * Raw-frame overlay for rational 3:5 scaling without lens distortion -- the reference's default pipeline, 1600x900
 * sensor frames -> 960x540 tiles (CameraManager(output_size=(540, 960)), cama/reproject.py:164,176-182,232-240).  There
 * cv2.remap's fixed-point taps repeat with period 3 destination / 5 source pixels (offsets 0,1,3; left/right weights
 * 32/0, 11/21, 21/11 of 32), so 12 destination pixels come from exactly 20 source pixels and every tap is a constant
 * byte offset: a band's source rows are streamed into LDS as one contiguous range, one thread per 12-pixel unit reads its
 * 2 x 60 bytes at a conflict-free odd dword stride.  Same bytes as cama_overlay_frames_raw / cama_resample_frames +
 * cama_overlay_frames.
 *   cama_raw35_plan      HOST pointers: mapx [C,W], mapy [C,H] (the separable float32 maps); out: vrows [C,H,2] uint32
 *                        ({top row | bottom row << 16, top weight | bottom weight << 8} per destination row; rows are
 *                        general, only the columns must follow the 3:5 pattern), band_rows [C,NB,2] int32 ({first
 *                        source row, number of source rows} of every band of R = cama_overlay_band_rows(W) rows; a band's
 *                        source rows are one contiguous range) and *max_src_rows.  Returns 1 when the kernel applies
 *                        (pattern verified for every column, W % 48 == 0, W0*3 % 16 == 0, all weighted taps inside the
 *                        frame, one 12-pixel unit per thread fits a workgroup), 0 when it does not (use
 *                        cama_overlay_frames_raw), < 0 on bad arguments.
 *   cama_overlay_frames_raw35   raw [F,C,H0,W0,3] uint8 (16-byte aligned), vrows / band_rows = the plan's tables on the
 *                        DEVICE; other arguments as cama_overlay_frames.
 */
int cama_raw35_plan(const float *mapx, const float *mapy, int32_t C, int32_t H, int32_t W, int32_t H0, int32_t W0,
                    uint32_t *vrows, int32_t *band_rows, int32_t *max_src_rows);
int cama_overlay_frames_raw35(const uint8_t *raw, int32_t H0, int32_t W0, const uint32_t *vrows, const int32_t *band_rows,
                              int32_t max_src_rows, uint8_t *mosaic, int64_t N, int32_t F, int32_t C, int32_t H, int32_t W,
                              int32_t cols, int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                              const void *scratch, size_t scratch_bytes, void *stream);

/*
 * Overlay half that reads RAW sensor frames: undistort + resize (the cv2.initUndistortRectifyMap + cv2.remap of
 * CameraManager.resize_image, cama/reproject.py:232-240) is fused into the overlay's source read, so the
 * reference-default pipeline 1600x900 JPEG frame -> 960x540 overlay -> 2880x1080 mosaic is one pass: every raw byte
 * is read once, every mosaic byte written once, no resized frame exists in memory.  Same semantics as
 * cama_resample_frames followed by cama_overlay_frames (tests/test_gpu_dropin.py checks byte equality).
 *   raw   [F,C,H0,W0,3] uint8      mapx, mapy: per camera c at mapx + c*(separable ? W : H*W), mapy + c*(separable ?
 *   H : H*W); separable as in cama_resample_frames.  Needs W % 16 == 0.  Other arguments as cama_overlay_frames.
 *   band_src_rows  NULL, or (separable maps only) device [C, NB, 2] int32 {first source row, number of source rows}
 *              that band b = rows [b*R, b*R+R) of camera c reads, R = cama_overlay_band_rows(W), NB = ceil(H/R),
 *              with max_src_rows their maximum, and
 *   tile_src_bytes device [C, tiles_x, 2] int32 {first source byte within a row (16-aligned), byte count (x16)} that
 *              column tile t = destination columns [t*W/tiles_x, (t+1)*W/tiles_x) of camera c reads (W/tiles_x a
 *              multiple of 16), max_tile_bytes their maximum byte count: together they enable the LDS-staged variant
 *              (each tile's source bytes streamed into LDS once with 16-byte loads, all taps from LDS) when
 *              W0*3 % 16 == 0 and a tile fits the 160 KB LDS; otherwise the gather variant runs
 */
int cama_overlay_frames_raw(const uint8_t *raw, int32_t H0, int32_t W0, const float *mapx, const float *mapy,
                            int32_t separable, const int32_t *band_src_rows, int32_t max_src_rows,
                            const int32_t *tile_src_bytes, int32_t tiles_x, int32_t max_tile_bytes,
                            uint8_t *mosaic, int64_t N, int32_t F, int32_t C,
                            int32_t H, int32_t W, int32_t cols,
                            int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                            const void *scratch, size_t scratch_bytes, void *stream);




/* How fast does the overlay run from `src` [F,C,H,W,3] into `mosaic` [F, rows*H, cols*W, 3] -- as a pure copy (no stamps), in
 * the XCD-contiguous order?  One untimed launch, then `reps` timed ones on `stream`; *ms_mean = their mean duration; blocks
 * until they are done; the mosaic ends up holding the plain mosaic of the frames.  For CHOOSING BUFFERS: the overlay's
 * bandwidth depends on where its source and destination sit physically relative to each other (about one destination
 * allocation in six runs at 0.83 of 8 TB/s, the others at 0.77, the same every time: profiles/r04_overlay_modes.txt section 5),
 * so a caller that keeps a mosaic (or frame) buffer for many launches allocates a few candidates, probes each and keeps the
 * fastest (cama_amd/engine.py: MosaicPool / Engine.place_frames).  The launches run under their own kernel
 * name (k_overlay_probe: the same code) so that a workload's kernel statistics keep them apart, and do not go through the
 * mapping table above.  W % 16 == 0 and 16-byte aligned buffers (CAMA_EINVAL otherwise).  (No reference counterpart.) */
int cama_overlay_probe(const uint8_t *src /* device */, uint8_t *mosaic /* device */, int32_t F, int32_t C, int32_t H, int32_t W,
                       int32_t cols, int32_t reps, double *ms_mean /* host */, void *stream);


/*
 * Stamp-only overlay for caller-supplied 2D points (the generic CameraManager.render_maps,
 * cama/reproject.py:246-257, for one image): points are (v,u) float64 in draw order.
 *   vu [n,2] float64, colour_id [n] uint8, image [H,W,3] uint8 updated in place.
 *   scratch: at least cama_stamp_scratch_bytes(H, W) bytes.
 */
size_t cama_stamp_scratch_bytes(int32_t H, int32_t W);
/*
 * EXTENSION (no reference semantics; the reference draws a disc per point and nothing between the points, SURVEY.md D1,
 * while BASELINE.json's north_star speaks of rasterised line segments): cama_stamp_points plus, for every point k with
 * link[k] != 0 (k > 0), a one-pixel-wide 8-connected Bresenham segment from point k - 1's truncated pixel to point k's
 * (both included; the integer error recurrence is stated in oracle_line_bresenham, oracle/cama_oracle.c), drawn under
 * point k's draw index and colour.  "Last writer wins" over discs and segments alike = per-pixel maximum of the draw
 * index.  link == NULL is cama_stamp_points.  Checked against the oracle's own restatement, never the default.
 */
int cama_stamp_polylines(const double *vu, const uint8_t *colour_id, const uint8_t *link, int64_t n,
                         uint8_t *image, int32_t H, int32_t W,
                         int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                         void *scratch, size_t scratch_bytes, void *stream);
/* The same with ANTI-ALIASED segments (Wu lines, coverages blended once per pixel): the one-image counterpart of
 * CAMA_BIN_SEGMENTS_WU, same definition (oracle/cama_oracle.c: oracle_render_frame_wu); n < 2^22.  (No reference counterpart.) */
int cama_stamp_polylines_wu(const double *vu, const uint8_t *colour_id, const uint8_t *link, int64_t n,
                         uint8_t *image, int32_t H, int32_t W,
                         int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                         void *scratch, size_t scratch_bytes, void *stream);
int cama_stamp_points(const double *vu, const uint8_t *colour_id, int64_t n,
                      uint8_t *image, int32_t H, int32_t W,
                      int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                      void *scratch, size_t scratch_bytes, void *stream);

/*
 * Per-clip static-map build on the device: MapManager.calculate_3d_instance_maps (cama/reproject.py:72-106, lift = 1)
 * and MapManager.load_3d_instance_maps (cama/reproject.py:42-70, lift = 0) for all labels at once, one thread per
 * densified point, float32 arithmetic in the reference's operation order (bit-identical, pinned by tests/golden).
 * The host supplies the segment table (which needs only the O(#label vertices) segment lengths):
 *   verts [V,2] float32; seg_v0/seg_num [S] for the segments with num = int(|seg| / solution) > 0;
 *   seg_off [S+1] exclusive scan of seg_num (seg_off[S] = N); seg_colour [S] palette index of the label's class
 *   raster [rows, cols] float32 (raster_is_f64 == 0) or float64: output is float64 exactly when the raster is
 *   solution / half_w / half_h / cx / cy: MapManager.solution, map_width/2, map_height/2, center_x, center_y as float32
 *   x, y, z [N] output vertex buffer (SoA), colour [N] uint8
 */
int cama_build_static_map(const float *verts, const int32_t *seg_v0, const int32_t *seg_num, const int64_t *seg_off,
                          const uint8_t *seg_colour, int32_t S, int64_t N, int32_t lift,
                          const void *raster, int32_t raster_is_f64, int32_t rows, int32_t cols,
                          float solution, float half_w, float half_h, float cx, float cy,
                          void *x, void *y, void *z, uint8_t *colour, void *stream);

/*
 * Undistort + resize resample of n frames of ONE camera through precomputed float32 maps: the
 * cv2.remap(image, mapx, mapy, INTER_LINEAR) of CameraManager.resize_image (cama/reproject.py:232-240); the
 * maps are what cv2.initUndistortRectifyMap(K_origin, d, None, K, (W,H), CV_32FC1) returns, built once per
 * camera on the host (the reference rebuilds them per frame).  OpenCV semantics restated: coordinates rounded
 * to 1/32 px, 15-bit fixed-point bilinear weights, BORDER_CONSTANT 0.  PARITY UNPINNED (no OpenCV on either box).
 *   src  frame i at src + i*src_stride_bytes, [H0,W0,3] uint8     dst frame i at dst + i*dst_stride_bytes, [H,W,3]
 *   mapx, mapy [H,W] float32 source coordinates of each destination pixel (separable == 0), or, for separable maps
 *              (zero distortion: mapx depends on the column only, mapy on the row only), mapx [W] and mapy [H]
 */
int cama_resample_frames(const uint8_t *src, int64_t src_stride_bytes, uint8_t *dst, int64_t dst_stride_bytes,
                         int32_t n, int32_t H0, int32_t W0, int32_t H, int32_t W,
                         const float *mapx, const float *mapy, int32_t separable, void *stream);

/* Host helper: half-widths of OpenCV's filled midpoint circle, hw[0..radius]; returns radius+1 or <0. */
int cama_circle_halfwidths(int32_t radius, int32_t *hw /* host */);

/* Rows per band the overlay kernel uses for images of width W (its LDS owner table is rows*W*4 bytes). */
int cama_overlay_band_rows(int32_t W);


/*
 * Device baseline-JPEG decode of a batch of camera frames: what CameraManager.read_resized_image /
 * read_image (cama/reproject.py:224,243: cv2.imread) and DatasetReader.yield_camera (cama/dataset_reader.py:72-76)
 * do with libjpeg(-turbo) on one host core per image.  Output is byte-identical to libjpeg-turbo's default decode
 * (islow IDCT, fancy upsampling; oracle/jpeg_oracle.py pins it to Pillow's bundled libjpeg-turbo).
 *
 * The host parses the markers (cama_amd/jpeg.py) and uploads: the bytes that hold the entropy-coded segments (whole
 * files are fine: a descriptor points at its segment, byte stuffing still in place, any alignment), one descriptor
 * per image, the Huffman table sets (device layout: cama_jpeg_huff_set_bytes() each, built by the host from the DHT
 * segments -- cama_amd/jpeg.py: build_huff_set is the reference builder, csrc/jpeg_kernels.hpp: JpegHuffRec the layout: 10-bit
 * symbol tables + per-length limits + a second level for the codes of 11..16 bits, then, since ABI 23, the state-transition
 * tables of the synchronisation phases, which depend on the table set alone and used to be derived by every workgroup)
 * and the quantisation tables ([set][component 0..2][64] uint16, natural order).
 * Scope: SOF0, 8 bit, 1 or 3 components in one interleaved scan, luma sampling 1x1 / 2x1 / 2x2 with 1x1 chroma,
 * restart intervals through per-interval descriptors.  Everything else is the caller's host fallback.
 *   cama_jpeg_plan    fills the derived descriptor fields and reports grid sizes and scratch bytes (host only)
 *   cama_jpeg_decode  stream: readable up to the next multiple of 4 bytes (it is read with aligned dword loads);
 *                     imgs_dev = device copy of the PLANNED descriptors; out: image `out_slot` of height*width*3 bytes,
 *                     out_stride bytes apart, BGR (bgr != 0: OpenCV order) or RGB; status [n] int32 on the device:
 *                     0 ok, != 0 the stream did not decode consistently (corrupt or unsupported): use the host
 *                     decoder for that image
 */
typedef struct cama_jpeg_image {
    uint64_t stream_off;        /* byte offset of the entropy-coded segment in `stream` (any alignment) */
    uint64_t clean_off;         /* [plan] byte offset of its unstuffed copy in the scratch, 16-byte aligned */
    uint64_t coef_off;          /* [plan] int16 elements into the coefficient scratch */
    uint64_t plane_off[3];      /* [plan] bytes into the plane scratch */
    uint32_t stream_len;        /* bytes up to (not including) the marker that ends the scan */
    uint32_t width, height;
    uint32_t ncomp;             /* 1 or 3 */
    uint32_t hs, vs;            /* luma sampling factors */
    uint32_t huff_set, quant_set;
    uint32_t comp_dc[3], comp_ac[3];   /* Huffman table selector (0/1) per component */
    uint32_t mx, my, bpm, total_blocks;                 /* [plan] MCU grid, blocks per MCU, blocks in the scan */
    uint32_t wg0, nwg, tile0, ntile;                    /* [plan] decode workgroups / unstuff tiles of this image */
    uint32_t plane_w[3], plane_h[3];                    /* [plan] padded component planes */
    /* Restart intervals (DRI): the image becomes one CAMA_JPEG_PIXELS descriptor (geometry, tables, planes, output;
     * stream_len = 0) followed by one CAMA_JPEG_SEGMENT descriptor per restart interval (its bytes between two RSTn
     * markers; width = MCUs in the interval * 8 * hs, height = 8 * vs, so that mx = MCU count, my = 1), which only
     * takes part in the entropy stages and writes into the parent's coefficients from block `first_block` on. */
    uint32_t kind;              /* CAMA_JPEG_WHOLE / _SEGMENT / _PIXELS */
    uint32_t parent;            /* _SEGMENT: index of its _PIXELS descriptor (earlier in the array) */
    uint32_t first_block;       /* _SEGMENT: first scan-order block of the interval */
    uint32_t out_slot;          /* _WHOLE / _PIXELS: image index in `out` */
} cama_jpeg_image;
#define CAMA_JPEG_WHOLE   0
#define CAMA_JPEG_SEGMENT 1
#define CAMA_JPEG_PIXELS  2

typedef struct cama_jpeg_plan_info {
    uint64_t scratch_bytes;
    uint32_t total_wgs, total_tiles, max_blocks, reserved;
} cama_jpeg_plan_info;

size_t cama_jpeg_image_bytes(void);       /* sizeof(cama_jpeg_image), for bindings that mirror the struct */
size_t cama_jpeg_huff_set_bytes(void);    /* bytes of one Huffman table set in device layout */
int cama_jpeg_plan(cama_jpeg_image *imgs /* host, in/out */, int32_t n, uint64_t stream_bytes,
                   cama_jpeg_plan_info *info /* host, out */);
/*
 * Restart-interval files (DRI): the byte positions of every RSTn marker (0xFF 0xD0..0xD7) in the uploaded bytes
 * [0, stream_bytes), found on the device before the descriptors exist (the reference's libjpeg walks them serially;
 * a Python-side search was 0.18 ms per 1600x900 image).  positions [capacity] uint32 and count [1] uint32 are device
 * memory; the list is UNORDERED, *count is the number of markers found even when it exceeds capacity (then only the
 * first `capacity` reservations were stored).  Hits outside the scans (file headers inside an uploaded span) are the
 * caller's to drop: it knows the scan ranges.  stream must be 16-byte aligned, stream_bytes < 2^32.
 */
int cama_jpeg_find_restarts(const uint8_t *stream, uint64_t stream_bytes, uint32_t *positions, uint32_t capacity,
                            uint32_t *count, void *stream_handle);
int cama_jpeg_decode(const uint8_t *stream, uint64_t stream_bytes, const cama_jpeg_image *imgs /* host, planned */,
                     const cama_jpeg_image *imgs_dev, int32_t n, const void *huff_sets, int32_t n_huff_sets,
                     const uint16_t *quant_sets, int32_t n_quant_sets, uint8_t *out, uint64_t out_stride, int32_t bgr,
                     void *scratch, size_t scratch_bytes, int32_t *status, void *stream_handle);

/*
 * Ingest helper (host only, no GPU work): read n files into caller-provided buffers -- the pinned arena the JPEG decoder
 * uploads from -- with `threads` native threads and no interpreter in the loop.  What cv2.imread does per image before it
 * decodes (cama/reproject.py:224,243), for a whole batch of camera frames at once: file i must be exactly sizes[i] bytes
 * (the caller took them from a directory scan); status[i] = 0 when dst[i] holds the whole file, 1 when the file could not be
 * opened, was shorter, or longer (the caller re-reads those the ordinary way).  Blocks until all files are read; bindings
 * should release their interpreter lock around the call (ctypes does).
 */
int cama_read_files(const char *const *paths /* host */, void *const *dst /* host */, const uint64_t *sizes /* host */,
                    int32_t n, int32_t threads, int32_t *status /* host */);

#ifdef __cplusplus
}
#endif
#endif /* CAMA_HIP_H */
