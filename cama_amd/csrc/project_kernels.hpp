// project_kernels.hpp -- part of libcama_hip.so (included by cama_hip.hip inside its anonymous namespace).
// Projection kernels: fp64 FMA chains, API-mode kernels (transform / crop / project), the fused per-frame
// kernels (coordinate-emitting mode and the count/fill stamp binning) and the bin scans.
#pragma once

// ------------------------------------------------------------------------------------------
// fp64 k-ordered FMA chains (see header: arithmetic contract)
// ------------------------------------------------------------------------------------------
typedef const double __attribute__((address_space(4))) kdouble;   // uniform loads from it are s_load (SGPR operands)
template <typename P>
__device__ __forceinline__ void affine3x4(P M, double x, double y, double z, double &ox, double &oy, double &oz)
{
    double a;
    a = M[0] * x; a = __builtin_fma(M[1], y, a); a = __builtin_fma(M[2], z, a);  a = __builtin_fma(M[3], 1.0, a);  ox = a;
    a = M[4] * x; a = __builtin_fma(M[5], y, a); a = __builtin_fma(M[6], z, a);  a = __builtin_fma(M[7], 1.0, a);  oy = a;
    a = M[8] * x; a = __builtin_fma(M[9], y, a); a = __builtin_fma(M[10], z, a); a = __builtin_fma(M[11], 1.0, a); oz = a;
}

template <typename P>
__device__ __forceinline__ void linear3x3(P K, double x, double y, double z, double &o0, double &o1, double &o2)
{
    double a;
    a = K[0] * x; a = __builtin_fma(K[1], y, a); a = __builtin_fma(K[2], z, a); o0 = a;
    a = K[3] * x; a = __builtin_fma(K[4], y, a); a = __builtin_fma(K[5], z, a); o1 = a;
    a = K[6] * x; a = __builtin_fma(K[7], y, a); a = __builtin_fma(K[8], z, a); o2 = a;
}

__device__ __forceinline__ bool in_crop(const Crop &c, double x, double y, double z)
{
    return (x >= c.v[0]) & (x <= c.v[1]) & (y >= c.v[2]) & (y <= c.v[3]) & (z >= c.v[4]) & (z <= c.v[5]);
}

// reproject.py:191-198.  h = K @ p_cam.  Visible iff h2 > 0 (mask_z), h2/h2 > 0 (false only for
// h2 = +inf, where the quotient is nan) and 0 <= u < W, 0 <= v < H on the IEEE quotients.
__device__ __forceinline__ bool pinhole(double h0, double h1, double h2, double Wd, double Hd,
                                        double &u, double &v)
{
    u = h0 / h2;
    v = h1 / h2;
    return (h2 > 0.0) & (h2 < __builtin_huge_val()) & (u >= 0.0) & (u < Wd) & (v >= 0.0) & (v < Hd);
}

// Bin-mode projection of one chassis-frame point into camera `m` (= 3x4 | 3x3): returns true and the packed
// truncated pixel iff the reference's mask (reproject.py:192-198) is true.  Same FMA chains as pinhole(); the two
// early-outs only skip work whose result is provably "not visible":
//  (a) K's third row is (0,0,k) for every pinhole K; then h2 = fma(k, pz, +-0) = k*pz, so the sign test can run on
//      the third affine row alone (5 fp64 ops instead of ~50 for the half-space behind the camera);
//  (b) h0 < -h2 or h0 > (W+1)*h2 (same for h1/H) puts the IEEE quotient below 0 / at or above W even after
//      rounding (one pixel of margin >> 1 ulp), so the two divisions (~28 fp64 ops) are skipped.
// m = chassis->camera 3x4 (row-major, 12 values), K = 3x3; P is `const double *` (LDS-staged) or a constant-address-
// space pointer (wave-uniform scalar loads: the matrices then live in SGPRs and cost no LDS or VMEM issue slots).
template <typename P>
__device__ __forceinline__ bool visible_pixel(P m, P K, double cx, double cy, double cz, double Wd, double Hd,
                                              uint32_t &uv)
{
    double a;
    a = m[8] * cx; a = __builtin_fma(m[9], cy, a); a = __builtin_fma(m[10], cz, a); a = __builtin_fma(m[11], 1.0, a);
    const double pz = a;
    const bool pinhole_row = (K[6] == 0.0) & (K[7] == 0.0);
    if (pinhole_row && !(K[8] * pz > 0.0)) return false;
    a = m[0] * cx; a = __builtin_fma(m[1], cy, a); a = __builtin_fma(m[2], cz, a); a = __builtin_fma(m[3], 1.0, a);
    const double px = a;
    a = m[4] * cx; a = __builtin_fma(m[5], cy, a); a = __builtin_fma(m[6], cz, a); a = __builtin_fma(m[7], 1.0, a);
    const double py = a;
    double h0, h1, h2;
    linear3x3(K, px, py, pz, h0, h1, h2);
    if (!(h2 > 0.0)) return false;
    if ((h0 < -h2) | (h0 > (Wd + 1.0) * h2) | (h1 < -h2) | (h1 > (Hd + 1.0) * h2)) return false;
    double u, v;
    if (!pinhole(h0, h1, h2, Wd, Hd, u, v)) return false;
    // reproject.py:249: astype(np.int32) truncation (values are >= 0 here)
    uv = (uint32_t)(int)u | ((uint32_t)(int)v << 16);
    return true;
}

// stage [C] x (3x4 chassis->camera | 3x3 K) into LDS
__device__ __forceinline__ void stage_cameras(double *s_cam, const double *c2cam, const double *K, int C)
{
    for (int t = threadIdx.x; t < C * CAM_STRIDE; t += BLOCK) {
        int c = t / CAM_STRIDE, k = t - c * CAM_STRIDE;
        s_cam[t] = (k < 12) ? c2cam[c * 16 + k] : K[c * 9 + (k - 12)];
    }
}

// ------------------------------------------------------------------------------------------
// API kernels (materialise coordinates)
// ------------------------------------------------------------------------------------------
// A workgroup's BLOCK points are 3 * BLOCK consecutive values of an [n, 3] array: moved with unit-stride loads / stores
// through LDS (a thread reading xyz[3 i], xyz[3 i + 1], xyz[3 i + 2] itself issues three stride-3 accesses).
template <typename T>
__device__ __forceinline__ void stage_points_in(const T *__restrict__ pts, int64_t i0, int64_t n, double *s_pts)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int64_t g = 3 * i0 + k * BLOCK + threadIdx.x;
        if (g < 3 * n) s_pts[k * BLOCK + threadIdx.x] = (double)pts[g];
    }
    __syncthreads();
}
__device__ __forceinline__ void stage_points_out(double *__restrict__ out, int64_t i0, int64_t n, const double *s_pts)
{
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int64_t g = 3 * i0 + k * BLOCK + threadIdx.x;
        if (g < 3 * n) out[g] = s_pts[k * BLOCK + threadIdx.x];
    }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_transform_points(const T *__restrict__ xyz, int64_t N,
                                                            const double *__restrict__ Tm, Crop crop,
                                                            int has_crop, double *__restrict__ out,
                                                            uint8_t *__restrict__ mask)
{
    __shared__ double s_m[12];
    __shared__ double s_pts[3 * BLOCK];
    const int f = blockIdx.y;
    if (threadIdx.x < 12) s_m[threadIdx.x] = Tm[(size_t)f * 16 + threadIdx.x];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK, i = i0 + threadIdx.x;
    stage_points_in(xyz, i0, N, s_pts);
    double ox = 0, oy = 0, oz = 0;
    if (i < N) {
        affine3x4((const double *)s_m, s_pts[3 * threadIdx.x], s_pts[3 * threadIdx.x + 1], s_pts[3 * threadIdx.x + 2], ox, oy, oz);
        if (mask) mask[(size_t)f * N + i] = has_crop ? (uint8_t)in_crop(crop, ox, oy, oz) : (uint8_t)1;
    }
    if (out) {                                              // (uniform; every thread rewrites the three values it read)
        s_pts[3 * threadIdx.x] = ox; s_pts[3 * threadIdx.x + 1] = oy; s_pts[3 * threadIdx.x + 2] = oz;
        stage_points_out(out + (size_t)f * N * 3, i0, N, s_pts);
    }
}

__global__ __launch_bounds__(BLOCK) void k_crop_points(const double *__restrict__ pts, int64_t n, Crop crop,
                                                       uint8_t *__restrict__ mask)
{
    __shared__ double s_pts[3 * BLOCK];
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK, i = i0 + threadIdx.x;
    stage_points_in(pts, i0, n, s_pts);
    if (i < n) mask[i] = (uint8_t)in_crop(crop, s_pts[3 * threadIdx.x], s_pts[3 * threadIdx.x + 1], s_pts[3 * threadIdx.x + 2]);
}

__global__ __launch_bounds__(BLOCK) void k_project_points(const double *__restrict__ pts, int64_t n,
                                                          const double *__restrict__ c2cam,
                                                          const double *__restrict__ K, int C, int W, int H,
                                                          double *__restrict__ vu, uint8_t *__restrict__ vis)
{
    __shared__ double s_cam[CAMA_MAX_CAMERAS * CAM_STRIDE];
    __shared__ double s_pts[3 * BLOCK];
    stage_cameras(s_cam, c2cam, K, C);
    const int64_t i0 = (int64_t)blockIdx.x * BLOCK, i = i0 + threadIdx.x;
    stage_points_in(pts, i0, n, s_pts);
    if (i >= n) return;
    const double x = s_pts[3 * threadIdx.x], y = s_pts[3 * threadIdx.x + 1], z = s_pts[3 * threadIdx.x + 2];
    const double Wd = (double)W, Hd = (double)H;
    for (int c = 0; c < C; ++c) {
        const double *m = s_cam + c * CAM_STRIDE;
        double px, py, pz, h0, h1, h2, u, v;
        affine3x4(m, x, y, z, px, py, pz);
        linear3x3(m + 12, px, py, pz, h0, h1, h2);
        const bool ok = pinhole(h0, h1, h2, Wd, Hd, u, v);
        reinterpret_cast<double2 *>(vu)[(size_t)c * n + i] = make_double2(v, u);
        vis[(size_t)c * n + i] = (uint8_t)ok;
    }
}

// ------------------------------------------------------------------------------------------
// fused per-frame kernel
// ------------------------------------------------------------------------------------------

// AABB of every 64 consecutive vertices (= what one wave of k_frames_project owns; cama_map_bounds).  NaN coordinates
// are ignored by fmin/fmax (such a vertex is never inside the crop box); an all-NaN run yields an empty box.
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_block_bounds(const void *x, const void *y, const void *z, int64_t N,
                                                        double *__restrict__ bounds)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const double inf = __builtin_huge_val();
    double v[6] = {inf, -inf, inf, -inf, inf, -inf};
    if (i < N) {
        v[0] = v[1] = (double)static_cast<const T *>(x)[i];
        v[2] = v[3] = (double)static_cast<const T *>(y)[i];
        v[4] = v[5] = (double)static_cast<const T *>(z)[i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (v[k] != v[k]) v[k] = (k & 1) ? -inf : inf;
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 6; k += 2) {
            v[k] = fmin(v[k], __shfl_xor(v[k], off, 64));
            v[k + 1] = fmax(v[k + 1], __shfl_xor(v[k + 1], off, 64));
        }
    }
    // one box per wave (64 consecutive vertices): lane k < 6 writes component k
    const int64_t sub = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (sub * 64 < N) {
        const uint32_t lane = threadIdx.x & 63u;
        const double r = lane == 0 ? v[0] : lane == 1 ? v[1] : lane == 2 ? v[2] : lane == 3 ? v[3] : lane == 4 ? v[4] : v[5];
        if (lane < 6u) bounds[(size_t)sub * 6 + lane] = r;
    }
}

// Multi-scene launches (cama_*_scenes): frame f of the launch belongs to scene f / frames_per_scene; what differs per
// scene -- vertex buffer, calibration, frames, mosaic -- comes from a device table read through wave-uniform (scalar)
// loads.  Everything in the scratch is indexed by the launch-wide frame number, so the scans, the scatter and the
// band lists are unaware of scenes.
struct SceneRef {
    const void *x, *y, *z;
    const uint8_t *colour;
    const uint32_t *key;
    const double *c2cam, *K;
    const uint8_t *src;
    uint8_t *mosaic;
    int64_t N;
};
static_assert(sizeof(SceneRef) == sizeof(cama_scene), "SceneRef mirrors cama_scene (include/cama_hip.h)");
typedef const SceneRef __attribute__((address_space(4))) kSceneRef;

struct FrameArgs {
    const SceneRef *scenes; // optional [S]: multi-scene launch (x .. K below are then placeholders, N = max over scenes)
    int frames_per_scene;
    const void *x, *y, *z;  // [N] each, float or double (template parameter T)
    const uint8_t *colour;
    const uint32_t *key;    // optional [N]: draw index << 1 | colour (maps stored in a different order than drawn)
    const double *bounds;   // optional [ceil(N/64),6]: AABB of every 64 consecutive vertices (k_block_bounds)
    const uint64_t *cam_mask;   // optional [F, vblocks]: 4 x 16 bits, cameras that may see wave w's 64 vertices of the
                                // block (k_block_cameras); 0 = outside the crop box / every frustum
    uint32_t vblocks;
    int64_t N;
    const double *w2c, *c2cam, *K;
    int C, W, H;
    Crop crop;
    // MODE_EMIT
    double *vu;
    uint8_t *vis, *crop_mask;
    // bin mode
    int band_shift, NB, radius;
    uint32_t nseg;          // segments (waves) per (frame, camera): ceil(N / BLOCK) * (BLOCK / 64)
    uint8_t *seg_cnt;       // [F*C*nseg] stamps in each segment (zeroed before k_frames_project)
    uint2 *stamps0;         // [F*C*nseg*64] compacted per-segment stamps
    uint32_t *counts, *cursor;
    const uint32_t *bin_off, *fc_base;
    uint2 *stamps;          // band-sorted stamps (k_stamps_scatter)
    int segments;           // EXTENSION (no reference semantics): records are 16 bytes {uv, key, uv of the polyline
                            // predecessor or ~0, 0}: the overlay also draws the one-pixel segment between the two
};

// Multi-scene launch: overwrite the per-scene fields of the (by-value) argument block with frame f's scene.
__device__ __forceinline__ void resolve_scene(FrameArgs &a, const int f)
{
    if (!a.scenes) return;
    kSceneRef *sc = (kSceneRef *)(a.scenes) + (f / a.frames_per_scene);
    a.x = sc->x; a.y = sc->y; a.z = sc->z;
    a.colour = sc->colour; a.key = sc->key;
    a.c2cam = sc->c2cam; a.K = sc->K;
    a.N = sc->N;
}

// Emit mode: materialise (v,u) + visibility for every (frame, camera, vertex).
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_frames_emit(FrameArgs a)
{
    __shared__ double s_w2c[12];
    __shared__ double s_cam[CAMA_MAX_CAMERAS * CAM_STRIDE];
    const int f = blockIdx.y;
    if (threadIdx.x < 12) s_w2c[threadIdx.x] = a.w2c[(size_t)f * 16 + threadIdx.x];
    stage_cameras(s_cam, a.c2cam, a.K, a.C);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.N) return;
    const double x = (double)static_cast<const T *>(a.x)[i], y = (double)static_cast<const T *>(a.y)[i],
                 z = (double)static_cast<const T *>(a.z)[i];
    double cx, cy, cz;
    affine3x4(s_w2c, x, y, z, cx, cy, cz);
    const bool in = in_crop(a.crop, cx, cy, cz);
    if (a.crop_mask) a.crop_mask[(size_t)f * a.N + i] = (uint8_t)in;
    const double Wd = (double)a.W, Hd = (double)a.H;
    for (int c = 0; c < a.C; ++c) {
        const size_t at = ((size_t)f * a.C + c) * a.N + i;
        bool ok = false;
        if (in) {
            const double *m = s_cam + c * CAM_STRIDE;
            double px, py, pz, h0, h1, h2, u, v;
            affine3x4(m, cx, cy, cz, px, py, pz);
            linear3x3(m + 12, px, py, pz, h0, h1, h2);
            ok = pinhole(h0, h1, h2, Wd, Hd, u, v);
            reinterpret_cast<double2 *>(a.vu)[at] = make_double2(v, u);
        }
        a.vis[at] = (uint8_t)ok;
    }
}

// Bin mode.  Every visible (vertex, camera) pair is a "stamp" {u:16, v:16, key = draw index << 1 | colour} that must
// reach the 1-2 row bands its disc touches.  The fp64 chain runs ONCE per (frame, vertex) (the kernel is fp64-VALU
// bound on dense maps: 74 % VALU-busy, profiles/r02_scalarcam_dense1e6_*; round 1 ran it twice, count + fill):
//   k_frames_project   projects, writes every wave's surviving stamps COMPACTED into that wave's fixed segment of
//                      (frame, camera) -- ballot + mbcnt, no atomics, no barrier -- with the count in a byte table,
//                      and accumulates the per-(frame, camera, band) stamp counts (per-workgroup LDS histogram, one
//                      global atomic per non-empty bin per workgroup);
//   k_scan_bands / k_scan_totals   exclusive scans of the counts;
//   k_stamps_scatter   re-reads only the 8-byte stamps (no geometry) and moves them to their band-sorted places.
constexpr int SEG = 64;                 // one segment = one wave's 64 vertices of one (frame, camera)

// One wave's share (64 consecutive vertices) of a vertex block in one frame: chassis-frame point, crop flag, draw key.
struct WaveVerts {
    double cx, cy, cz;
    uint32_t key;
    bool in;
    uint32_t cams;          // cameras that may see these 64 vertices (wave-uniform); 0 = nothing to do
    uint32_t seg;           // segment index inside (frame, camera)
    // segment extension: is this vertex joined to its predecessor on the polyline (bit 1 of its colour byte)?  Lane 0's
    // predecessor belongs to the previous wave: its chassis point travels in (hx, hy, hz, hin) of lane 0
    bool link;
    double hx, hy, hz;
    bool hin;
};

template <typename T>
__device__ __forceinline__ WaveVerts load_wave_verts(const FrameArgs &a, const int64_t vblock, const int f, const uint64_t cams4,
                                                     const uint32_t seg_base)
{
    WaveVerts v{0.0, 0.0, 0.0, 0u, false, 0u, 0u, false, 0.0, 0.0, 0.0, false};
    // this wave's 16 camera bits (wave-uniform: kept in an SGPR)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    v.cams = (uint32_t)(cams4 >> (16u * wave)) & 0xffffu;
    v.seg = seg_base + wave;            // (vblock * 4, or -- planned launches -- the block's rank among its frame's survivors * 4)
    if (!v.cams) return v;
    // world->chassis + crop FIRST, with the frame's matrix read through wave-uniform (scalar) loads: on site-sized
    // maps ~95 % of the waves end here
    const double *w2c = a.w2c + (size_t)f * 16;
    const int64_t i = vblock * BLOCK + threadIdx.x;
    if (i < a.N) {
        const double x = (double)static_cast<const T *>(a.x)[i], y = (double)static_cast<const T *>(a.y)[i],
                     z = (double)static_cast<const T *>(a.z)[i];
        affine3x4(w2c, x, y, z, v.cx, v.cy, v.cz);
        v.in = in_crop(a.crop, v.cx, v.cy, v.cz);
        // draw index << 1 | colour.  Spatially re-ordered maps carry it per vertex (a.key), otherwise it is the
        // storage index itself
        const uint32_t cbyte = a.key ? 0u : (uint32_t)a.colour[i];
        v.key = a.key ? a.key[i] : (((uint32_t)i << 1) | (cbyte & 1u));
        if (a.segments) {
            v.link = v.in && i > 0 && ((cbyte >> 1) & 1u);
            if ((threadIdx.x & 63u) == 0u && v.link) {          // the predecessor lives in the previous wave: fetch it here
                const double px = (double)static_cast<const T *>(a.x)[i - 1], py = (double)static_cast<const T *>(a.y)[i - 1],
                             pz = (double)static_cast<const T *>(a.z)[i - 1];
                affine3x4(w2c, px, py, pz, v.hx, v.hy, v.hz);
                v.hin = in_crop(a.crop, v.hx, v.hy, v.hz);
            }
        }
    }
    // whole wave outside the crop box: done (its segments stay empty: the count table was zeroed)
    if (!__ballot(v.in)) v.cams = 0;
    return v;
}

// Rows a segment record can touch: its disc (vi +- radius) and its segment (vi .. vp); the anti-aliased variant (a.segments == 2,
// Wu) writes the row above and below the exact line too, one row more at either end.
__device__ __forceinline__ void record_bands(const FrameArgs &a, const int vi, const int vp, const bool has_prev, int &b0, int &b1)
{
    const int e = (a.segments == 2 && has_prev) ? 1 : 0;
    const int r = max(a.radius, e);
    b0 = max(min(vi - r, vp - e), 0) >> a.band_shift;
    b1 = min(max(vi + r, vp + e), a.H - 1) >> a.band_shift;
}

// The wave's stamps of camera c (uv = packed truncated pixel or 0xffffffff): drop same-pixel predecessors, compact into the
// wave's segment, count, add the bands to the LDS histogram.
// Segment extension: the same with 16-byte records {uv, key, uv of the predecessor | ~0, 0}.  `uv_halo` = lane 0's
// predecessor pixel in this camera (or ~0).  A record reaches every band its disc rows OR its segment rows touch.
__device__ __forceinline__ void emit_wave_segments(const FrameArgs &a, const int f, const int c, const WaveVerts &v,
                                                   const uint32_t uv, const uint32_t uv_halo, uint32_t *s_cnt)
{
    const uint32_t lane = __lane_id();
    uint32_t uv_prev = __shfl_up(uv, 1, 64);
    if (lane == 0u) uv_prev = uv_halo;
    // joined: both ends visible in this camera, neighbours on the polyline, and not the same pixel (the disc covers that)
    const bool has_seg = v.link && uv != 0xffffffffu && uv_prev != 0xffffffffu && uv_prev != uv;
    const uint32_t uv_next = __shfl_down(uv, 1, 64);
    const uint32_t key_next = __shfl_down(v.key, 1, 64);
    // (a disc under a later point's identical disc is invisible -- but its segment is not)
    const bool covered = (lane != 63u) && (uv_next == uv) && (key_next > v.key) && !has_seg;
    const bool keep = (uv != 0xffffffffu) && !covered;
    const uint64_t m = __ballot(keep);
    if (!m) return;
    const size_t fcseg = ((size_t)f * a.C + c) * a.nseg + v.seg;
    if (keep) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        reinterpret_cast<uint4 *>(a.stamps0)[fcseg * SEG + rank] = make_uint4(uv, v.key, has_seg ? uv_prev : 0xffffffffu, 0u);
        const int vi = (int)(uv >> 16), vp = has_seg ? (int)(uv_prev >> 16) : vi;
        int b0, b1;
        record_bands(a, vi, vp, has_seg, b0, b1);
        for (int b = b0; b <= b1; ++b) atomicAdd(&s_cnt[c * a.NB + b], 1u);
    }
    if (lane == 0u) a.seg_cnt[fcseg] = (uint8_t)__popcll(m);
}

// How far ahead in its wave a stamp looks for a later point on the same pixel (see emit_wave_stamps)
constexpr int DEDUP_WINDOW = 8;

__device__ __forceinline__ void emit_wave_stamps(const FrameArgs &a, const int f, const int c, const WaveVerts &v,
                                                 const uint32_t uv, uint32_t *s_cnt)
{
    const uint32_t lane = __lane_id();
    const bool valid = uv != 0xffffffffu;
    const uint64_t mv = __ballot(valid);
    if (!mv) return;                    // wave-uniform
    // A disc is invisible if a LATER point (higher draw index) stamps the very same pixel (same footprint).  The
    // following lanes are the following vertices of the polyline, so on dense maps (1 cm spacing) most far-range stamps
    // collapse here, exactly, before they cost HBM or LDS traffic.
    // Round 5: not just the next lane but the next DEDUP_WINDOW -- where a polyline crosses the image at a shallow angle far
    // from the camera its points hop back and forth between two or three pixels (A B A B ..., A A B A B B ...), which the
    // next-lane test cannot see: on the 10^6-vertex site maps 36 % of the stamps that survived it were still duplicates of
    // a later stamp; a window of 8 leaves 10 % (the whole wave: 8 %).  The window costs ~40 instructions per (wave, camera)
    // -- +23 % on the dense lane map's projection when every wave paid it, for 0.8 % fewer stamps: lanes that run towards
    // the vanishing point step monotonically, A A A B B B, and the next-lane test gets them all -- so a wave runs it only
    // when it shows the pattern: some lane's pixel comes back two lanes on with another pixel in between (A B A).  The test
    // is wave-uniform; which stamps are dropped is a matter of speed, never of pixels.
    const uint32_t uv_1 = __shfl_down(uv, 1, 64), uv_2 = __shfl_down(uv, 2, 64);
    const bool monotone_keys = a.key == nullptr;                     // draw order = buffer order: later lanes carry greater keys
    const uint32_t key_1 = monotone_keys ? 0xffffffffu : __shfl_down(v.key, 1, 64);
    bool covered = valid && (lane < 63u) && (uv_1 == uv) && (key_1 > v.key);
    if (__ballot(valid && (lane < 62u) && (uv_2 == uv) && (uv_1 != uv))) {       // wave-uniform
        const uint32_t key_2 = monotone_keys ? 0xffffffffu : __shfl_down(v.key, 2, 64);
        covered |= valid && (lane < 62u) && (uv_2 == uv) && (key_2 > v.key);
#pragma unroll
        for (int d = 3; d <= DEDUP_WINDOW; ++d) {
            const uint32_t uv_d = __shfl_down(uv, d, 64);
            const uint32_t key_d = monotone_keys ? 0xffffffffu : __shfl_down(v.key, d, 64);
            covered |= valid && (lane + (uint32_t)d < 64u) && (uv_d == uv) && (key_d > v.key);
        }
    }
    const bool keep = valid && !covered;
    const uint64_t m = __ballot(keep);
    if (!m) return;                     // wave-uniform
    const size_t fcseg = ((size_t)f * a.C + c) * a.nseg + v.seg;
    if (keep) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        a.stamps0[fcseg * SEG + rank] = make_uint2(uv, v.key);
        // band histogram: per-lane LDS atomics (counting the wave's stamps per distinct band by ballot first was
        // measured slower on the dense map, 475 -> 527 us: same-address ds_add is cheaper than the leader loop)
        const int vi = (int)(uv >> 16);
        const int b0 = max(vi - a.radius, 0) >> a.band_shift;
        const int b1 = min(vi + a.radius, a.H - 1) >> a.band_shift;       // b1 <= b0 + 1 (host checks 2r <= R)
        const uint32_t l0 = (uint32_t)(c * a.NB + b0);
        atomicAdd(&s_cnt[l0], 1u);
        if (b1 != b0) atomicAdd(&s_cnt[l0 + 1], 1u);
    }
    if (lane == 0u) a.seg_cnt[fcseg] = (uint8_t)__popcll(m);
}

// Vertex block `vblock` of frame f: every wave projects its 64 vertices, compacts the surviving stamps into its segments
// and adds their bands to the workgroup's LDS histogram s_cnt [C*NB] (zeroed by the caller).  No barrier inside: the
// waves of a workgroup run through their blocks independently; the caller flushes s_cnt after a barrier.
template <typename T>
__device__ __forceinline__ void project_block(const FrameArgs &a, const int64_t vblock, const int f, const uint64_t cams4,
                                              uint32_t *s_cnt, const uint32_t seg_base)
{
    const WaveVerts v = load_wave_verts<T>(a, vblock, f, cams4, seg_base);
    if (!v.cams) return;
    const double Wd = (double)a.W, Hd = (double)a.H;
    // only the cameras the wave's mask lets through (wave-uniform scalar loop: the kernel issues as many SALU as VALU
    // instructions, PMC: profiles/r02_project_dense1e6_pmc_sq.csv)
    for (uint32_t todo = v.cams & ((1u << a.C) - 1u); todo; todo &= todo - 1u) {
        const int c = __builtin_ctz(todo);
        uint32_t uv = 0xffffffffu;      // packed truncated pixel, or "not visible"
        if (v.in) {
            uint32_t packed;
            // the camera's 21 doubles through wave-uniform scalar loads (constant address space): read as LDS
            // broadcasts they were ~126 ds_read_b64 per lane and made dense maps LDS-issue bound
            if (visible_pixel<kdouble *>((kdouble *)(a.c2cam + (size_t)c * 16), (kdouble *)(a.K + (size_t)c * 9), v.cx,
                                         v.cy, v.cz, Wd, Hd, packed))
                uv = packed;
        }
        if (a.segments) {               // (wave-uniform; the halo chain runs for lane 0 only -- an opt-in extension pays it)
            uint32_t uv_halo = 0xffffffffu;
            if (v.link && v.hin && (threadIdx.x & 63u) == 0u) {
                uint32_t packed;
                if (visible_pixel<kdouble *>((kdouble *)(a.c2cam + (size_t)c * 16), (kdouble *)(a.K + (size_t)c * 9), v.hx,
                                             v.hy, v.hz, Wd, Hd, packed))
                    uv_halo = packed;
            }
            emit_wave_segments(a, f, c, v, uv, uv_halo, s_cnt);
        } else
            emit_wave_stamps(a, f, c, v, uv, s_cnt);
    }
}

// workgroup-level frame of project_block: zero the histogram, run `body`, add the non-empty bins to the frame's counts
__device__ __forceinline__ void hist_clear(const FrameArgs &a, uint32_t *s_cnt)
{
    for (int t = threadIdx.x; t < a.C * a.NB; t += BLOCK) s_cnt[t] = 0u;
    __syncthreads();
}
__device__ __forceinline__ void hist_flush(const FrameArgs &a, const int f, const uint32_t *s_cnt)
{
    const int nloc = a.C * a.NB;
    __syncthreads();
    for (int t = threadIdx.x; t < nloc; t += BLOCK) {
        const uint32_t n = s_cnt[t];
        if (n) atomicAdd(&a.counts[(size_t)f * nloc + t], n);
    }
}

// grid (ceil(vblocks / vb_per_wg) padded to a multiple of 8, F): a workgroup runs vb_per_wg consecutive vertex blocks of
// one frame.  One block per workgroup made big maps dispatch-bound: on the dense 1e6-vertex map (156 k workgroups per
// launch) an ablation with the camera loop removed still took 278 of 534 us -- workgroup launch, histogram clear /
// flush and two barriers per 256 vertices (profiles/r02_project_ablation.txt).
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_frames_project(FrameArgs a, const int vb_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];  // [C*NB] stamp counts of this workgroup
    const int f = blockIdx.y;
    resolve_scene(a, f);
    const int64_t vb0 = (int64_t)blockIdx.x * vb_per_wg;
    if (vb0 * BLOCK >= a.N) return;
    hist_clear(a, s_hist);
    // with the map's block AABBs (cama_map_bounds -> k_block_cameras) a block outside the crop box or every frustum is
    // not even read; the others know, per wave, which cameras can see them at all
    const uint64_t *cm = a.cam_mask ? a.cam_mask + (size_t)f * a.vblocks : nullptr;
    // (two blocks per iteration -- two independent fp64 chains per lane -- was measured: 266 -> 280 us, no gain from ILP)
    for (int b = 0; b < vb_per_wg; ++b) {
        const int64_t vb = vb0 + b;
        if (vb * BLOCK >= a.N) break;
        const uint64_t cams = cm ? cm[vb] : ~0ull;                           // scalar load
        if (cams) project_block<T>(a, vb, f, cams, s_hist, (uint32_t)vb * (BLOCK / SEG));
    }
    hist_flush(a, f, s_hist);
}

// Band sort of the compacted stamps: one workgroup per SCATTER_SEGS consecutive segments of one (frame, camera).  Work
// is proportional to the stamps that exist (a prefix sum of the segment counts maps thread -> stamp), every stamp is
// ranked inside the workgroup with an LDS atomic on its band's counter, the workgroup reserves its share of each
// non-empty band with one global atomic, and the stamps go to bin_off + reserved base + rank.
#ifndef SCATTER_SEGS_N
#define SCATTER_SEGS_N 256
#endif
constexpr int SCATTER_SEGS = SCATTER_SEGS_N;   // 64, 128 or 256 (<= BLOCK): fewer, fatter workgroups = fewer reservations
constexpr int SCATTER_K = 4;            // stamps per thread per round (register-resident between rank and write)
static_assert(SCATTER_SEGS <= BLOCK && (SCATTER_SEGS & (SCATTER_SEGS - 1)) == 0 && SCATTER_SEGS >= 64, "SCATTER_SEGS");

__global__ __launch_bounds__(BLOCK) void k_stamps_scatter(FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];  // [NB] counts | [NB] bases
    __shared__ uint32_t s_off[SCATTER_SEGS + 1];
    __shared__ uint32_t s_wave[BLOCK / 64];
    const uint32_t fc = blockIdx.y, seg0 = blockIdx.x * SCATTER_SEGS;
    const int NB = a.NB;
    uint32_t *s_cnt = s_hist, *s_base = s_hist + NB;
    {
        // inclusive scan of the SCATTER_SEGS segment counts: inside each wave by shuffles, across waves through LDS
        const uint32_t sg = seg0 + threadIdx.x;
        const uint32_t v = (threadIdx.x < SCATTER_SEGS && sg < a.nseg) ? (uint32_t)a.seg_cnt[(size_t)fc * a.nseg + sg] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(x, d, 64);
            if ((int)(threadIdx.x & 63) >= d) x += t;
        }
        if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = x;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += s_wave[w];
        if (threadIdx.x < SCATTER_SEGS) s_off[threadIdx.x + 1] = before + x;
        if (threadIdx.x == 0) s_off[0] = 0u;
    }
    for (int t = threadIdx.x; t < NB; t += BLOCK) s_cnt[t] = 0u;
    __syncthreads();
    const uint32_t total = s_off[SCATTER_SEGS];
    if (!total) return;
    const uint2 *segs = a.stamps0 + ((size_t)fc * a.nseg + seg0) * SEG;
    const size_t gbin0 = (size_t)fc * NB;
    const uint32_t fcb = a.fc_base[fc];
    for (uint32_t base = 0; base < total; base += BLOCK * SCATTER_K) {
        uint32_t e_uv[SCATTER_K], e_key[SCATTER_K], e_slot[2 * SCATTER_K];
#pragma unroll
        for (int j = 0; j < SCATTER_K; ++j) {
            e_slot[2 * j] = e_slot[2 * j + 1] = 0xffffffffu;
            e_uv[j] = e_key[j] = 0u;
            const uint32_t t = base + j * BLOCK + threadIdx.x;
            if (t < total) {
                uint32_t lo = 0;        // the segment that holds stamp t: largest lo with s_off[lo] <= t
#pragma unroll
                for (uint32_t step = SCATTER_SEGS / 2; step; step >>= 1)
                    if (s_off[lo + step] <= t) lo += step;
                const uint2 r = segs[(size_t)lo * SEG + (t - s_off[lo])];
                const int vi = (int)(r.x >> 16);
                const int b0 = max(vi - a.radius, 0) >> a.band_shift;
                const int b1 = min(vi + a.radius, a.H - 1) >> a.band_shift;
                e_uv[j] = r.x;
                e_key[j] = r.y;
                e_slot[2 * j] = ((uint32_t)b0 << 12) | atomicAdd(&s_cnt[b0], 1u);         // rank < BLOCK * K * 2 <= 4096
                if (b1 != b0) e_slot[2 * j + 1] = ((uint32_t)b1 << 12) | atomicAdd(&s_cnt[b1], 1u);
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < NB; t += BLOCK) {
            const uint32_t n = s_cnt[t];
            uint32_t bs = 0;
            if (n) {
                bs = atomicAdd(&a.cursor[gbin0 + t], n) + a.bin_off[gbin0 + t] + fcb;
                s_cnt[t] = 0u;
            }
            s_base[t] = bs;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2 * SCATTER_K; ++j)
            if (e_slot[j] != 0xffffffffu)
                a.stamps[(size_t)s_base[e_slot[j] >> 12] + (e_slot[j] & 0xfffu)] = make_uint2(e_uv[j >> 1], e_key[j >> 1]);
        __syncthreads();                 // s_base / s_cnt are rewritten by the next round
    }
}

// The same for 16-byte segment records (EXTENSION): a record goes to EVERY band between the first row its disc or its segment
// touches and the last -- usually one or two, a whole image column of them for a segment that crosses the frame -- so ranks
// are not kept in registers: a counting sweep, one reservation per non-empty band, then a second sweep that ranks and writes.
__global__ __launch_bounds__(BLOCK) void k_stamps_scatter_seg(FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];  // [NB] counts | [NB] bases
    __shared__ uint32_t s_off[SCATTER_SEGS + 1];
    __shared__ uint32_t s_wave[BLOCK / 64];
    const uint32_t fc = blockIdx.y, seg0 = blockIdx.x * SCATTER_SEGS;
    const int NB = a.NB;
    uint32_t *s_cnt = s_hist, *s_base = s_hist + NB;
    {
        const uint32_t sg = seg0 + threadIdx.x;
        const uint32_t v = (threadIdx.x < SCATTER_SEGS && sg < a.nseg) ? (uint32_t)a.seg_cnt[(size_t)fc * a.nseg + sg] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(x, d, 64);
            if ((int)(threadIdx.x & 63) >= d) x += t;
        }
        if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = x;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += s_wave[w];
        if (threadIdx.x < SCATTER_SEGS) s_off[threadIdx.x + 1] = before + x;
        if (threadIdx.x == 0) s_off[0] = 0u;
    }
    for (int t = threadIdx.x; t < NB; t += BLOCK) s_cnt[t] = 0u;
    __syncthreads();
    const uint32_t total = s_off[SCATTER_SEGS];
    if (!total) return;
    const uint4 *segs = reinterpret_cast<const uint4 *>(a.stamps0) + ((size_t)fc * a.nseg + seg0) * SEG;
    uint4 *sorted = reinterpret_cast<uint4 *>(a.stamps);
    const size_t gbin0 = (size_t)fc * NB;
    const uint32_t fcb = a.fc_base[fc];
    for (uint32_t base = 0; base < total; base += BLOCK) {
        const uint32_t t = base + threadIdx.x;
        uint4 r = make_uint4(0, 0, 0, 0);
        int b0 = 0, b1 = -1;
        if (t < total) {
            uint32_t lo = 0;
#pragma unroll
            for (uint32_t step = SCATTER_SEGS / 2; step; step >>= 1)
                if (s_off[lo + step] <= t) lo += step;
            r = segs[(size_t)lo * SEG + (t - s_off[lo])];
            const int vi = (int)(r.x >> 16), vp = r.z != 0xffffffffu ? (int)(r.z >> 16) : vi;
            record_bands(a, vi, vp, r.z != 0xffffffffu, b0, b1);
            for (int b = b0; b <= b1; ++b) atomicAdd(&s_cnt[b], 1u);
        }
        __syncthreads();
        for (int b = threadIdx.x; b < NB; b += BLOCK) {
            const uint32_t n = s_cnt[b];
            s_base[b] = n ? atomicAdd(&a.cursor[gbin0 + b], n) + a.bin_off[gbin0 + b] + fcb : 0u;
            s_cnt[b] = 0u;
        }
        __syncthreads();
        for (int b = b0; b <= b1; ++b) sorted[(size_t)s_base[b] + atomicAdd(&s_cnt[b], 1u)] = r;
        __syncthreads();
        for (int b = threadIdx.x; b < NB; b += BLOCK) s_cnt[b] = 0u;
        __syncthreads();
    }
}

// Which cameras can a vertex block reach?  One THREAD per (box, frame), one box per WAVE of a vertex block (64
// consecutive vertices: on the dense 1e6-vertex lane map a 256-vertex box left 2.14 cameras per block where 0.98 have a
// visible vertex -- close to the car a 2.5 m box spans two or three frusta): each world AABB becomes a
// conservative chassis-frame box (centre +- sum |m_k| * half extent, widened by a margin 7 orders above fp64 rounding),
// clipped to the crop box
// (vertices outside it are dropped before projection), and that box is tested against the five half-spaces every visible
// point satisfies in homogeneous image coordinates h = K (R p + t):
//     h2 > 0,   h0 >= 0,   h0 - W h2 < 0,   h1 >= 0,   h1 - H h2 < 0
// (each a linear functional of the chassis point; a box wholly on the wrong side of any one of them has no visible point in
// that camera -- for points with h2 <= 0 nothing is visible anyway, so the sign arguments only need h2 > 0).  Margins of
// 1e-9 relative + 1e-9 absolute are >= 6 orders above fp64 rounding of the exact chain, and every comparison is false on
// NaN, i.e. "may be visible": the mask never removes a stamp, the output is bit-identical with and without it.
// Bit 16 w + c of cam_mask[f * vblocks + b] = camera c may see wave w's vertices of block b in frame f; 0 = the block is
// outside the crop box too.
// The projection kernel then skips the whole fp64 chain of the other cameras (wave-uniform branch): on dense lane maps a
// block is in view of 1-2 of the 6 cameras.  With WORKLIST the surviving (block, frame) items are also appended to 8
// per-XCD work lists for persistent workgroups (site-sized maps: ~95 % of the items are outside the crop box, and even an
// empty workgroup costs ~0.8 ns of dispatch): vertex block b always goes to list b % 8 and list l is walked by the
// workgroups with blockIdx.x % 8 == l, i.e. by XCD l, whose L2 keeps its eighth of the live vertex blocks.
__device__ __forceinline__ bool box_beyond(const double n0, const double n1, const double n2, const double d,
                                           const double *mid, const double *rad, const bool want_positive)
{
    // functional L(p) = n.p + d over the box mid +- rad: returns true iff L < 0 on the whole box (want_positive: the
    // visible side is L >= 0 / L > 0) or L > 0 on the whole box (!want_positive: the visible side is L < 0), with margin
    const double v = n0 * mid[0] + n1 * mid[1] + n2 * mid[2] + d;
    const double s = fabs(n0) * rad[0] + fabs(n1) * rad[1] + fabs(n2) * rad[2];
    const double mag = fabs(n0) * (fabs(mid[0]) + rad[0]) + fabs(n1) * (fabs(mid[1]) + rad[1]) +
                       fabs(n2) * (fabs(mid[2]) + rad[2]) + fabs(d);
    const double margin = 1e-9 + 1e-9 * mag;
    return want_positive ? (v + s + margin < 0.0) : (v - s - margin > 0.0);
}

// Cameras that may see any vertex of the world AABB (centre w, half extents e) in the frame with world->chassis matrix m.
__device__ __forceinline__ uint32_t box_cameras(const double *m, const double wx, const double wy, const double wz,
                                                const double ex, const double ey, const double ez, const Crop &crop,
                                                const double *fn, const int C)
{
    uint32_t mask = 0;
    double mid[3], rad[3];
    bool outside = false, finite = true;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double c0 = m[4 * r], c1 = m[4 * r + 1], c2 = m[4 * r + 2], c3 = m[4 * r + 3];
        const double ctr = c0 * wx + c1 * wy + c2 * wz + c3;
        const double rd = fabs(c0) * ex + fabs(c1) * ey + fabs(c2) * ez;
        const double mag = fabs(c0 * wx) + fabs(c1 * wy) + fabs(c2 * wz) + fabs(c3) + rd;
        const double margin = 1e-6 + 1e-9 * mag;
        double lo = ctr - rd - margin, hi = ctr + rd + margin;
        outside |= (hi < crop.v[2 * r]) | (lo > crop.v[2 * r + 1]);
        finite &= (lo == lo) & (hi == hi) & (fabs(lo) < 1e300) & (fabs(hi) < 1e300);
        lo = fmax(lo, crop.v[2 * r]);                   // clip to the crop box: only in-crop vertices are projected
        hi = fmin(hi, crop.v[2 * r + 1]);
        mid[r] = 0.5 * (lo + hi);
        rad[r] = 0.5 * (hi - lo);
    }
    if (!finite) {
        mask = (1u << C) - 1u;                          // NaN / inf anywhere: no claim, keep every camera
    } else if (!outside) {
        for (int c = 0; c < C; ++c) {
            kdouble *P = (kdouble *)(fn + (size_t)c * 20);      // wave-uniform: scalar loads
            bool gone = box_beyond(P[8], P[9], P[10], P[11], mid, rad, true);                               // h2 > 0
            gone |= box_beyond(P[0], P[1], P[2], P[3], mid, rad, true);                                     // h0 >= 0
            gone |= box_beyond(P[4], P[5], P[6], P[7], mid, rad, true);                                     // h1 >= 0
            gone |= box_beyond(P[12], P[13], P[14], P[15], mid, rad, false);                                // h0 < W h2
            gone |= box_beyond(P[16], P[17], P[18], P[19], mid, rad, false);                                // h1 < H h2
            if (!gone) mask |= 1u << c;
        }
    }
    return mask;
}

// The five functionals of every camera, once per launch: rows h0, h1, h2 of K [R | t], then h0 - W h2, h1 - H h2.
// fn [C][5][4]; k_block_cameras reads them through scalar loads (as LDS broadcasts they were 120 ds_read_b64 per thread).
// Workgroup 0 does that; workgroups 1.. (launched only for the candidate pre-pass below) write, per frame, a WORLD-space
// AABB that contains every point the frame's world->chassis matrix can take into the crop box: the pre-image of the crop
// box under c = A w + t is A^-1 (crop - t), bounded per world axis by centre +- sum |A^-1| half-extents.  A^-1 is the
// adjugate over the determinant and is only trusted when A A^-1 = I to 1e-12 (rigid poses give ~1e-16); the margin
// (1e-6 + 1e-9 of the magnitudes involved) is >= 3 orders above the rounding of the per-vertex chain it must cover.
// Anything not finite, or a matrix that fails the check: the box is (-inf, +inf), i.e. no claim.
__global__ void k_camera_functionals(const double *__restrict__ c2cam, const double *__restrict__ K, int C, int W, int H,
                                     double *__restrict__ fn, const double *__restrict__ w2c, uint32_t F, Crop crop,
                                     double *__restrict__ frame_box)
{
    if (blockIdx.x == 0) {
        if ((int)threadIdx.x >= C * 20) return;
        const int c = threadIdx.x / 20, i = (threadIdx.x % 20) / 4, j = threadIdx.x % 4;
        const double *M = c2cam + (size_t)c * 16, *Kc = K + (size_t)c * 9;
        const auto row = [&](int r) { return Kc[3 * r] * M[j] + Kc[3 * r + 1] * M[4 + j] + Kc[3 * r + 2] * M[8 + j]; };
        fn[threadIdx.x] = i < 3 ? row(i) : row(i - 3) - (i == 3 ? (double)W : (double)H) * row(2);
        return;
    }
    const uint32_t f = (blockIdx.x - 1u) * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const double *m = w2c + (size_t)f * 16;
    double a[3][3], t[3], inv[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        a[r][0] = m[4 * r]; a[r][1] = m[4 * r + 1]; a[r][2] = m[4 * r + 2]; t[r] = m[4 * r + 3];
    }
    // inv = adj(A) / det
    inv[0][0] = a[1][1] * a[2][2] - a[1][2] * a[2][1];
    inv[0][1] = a[0][2] * a[2][1] - a[0][1] * a[2][2];
    inv[0][2] = a[0][1] * a[1][2] - a[0][2] * a[1][1];
    inv[1][0] = a[1][2] * a[2][0] - a[1][0] * a[2][2];
    inv[1][1] = a[0][0] * a[2][2] - a[0][2] * a[2][0];
    inv[1][2] = a[0][2] * a[1][0] - a[0][0] * a[1][2];
    inv[2][0] = a[1][0] * a[2][1] - a[1][1] * a[2][0];
    inv[2][1] = a[0][1] * a[2][0] - a[0][0] * a[2][1];
    inv[2][2] = a[0][0] * a[1][1] - a[0][1] * a[1][0];
    const double det = a[0][0] * inv[0][0] + a[0][1] * inv[1][0] + a[0][2] * inv[2][0];
    bool ok = fabs(det) > 0.0 && fabs(det) < 1e300;                           // false on NaN
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) inv[i][j] /= det;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double r = a[i][0] * inv[0][j] + a[i][1] * inv[1][j] + a[i][2] * inv[2][j] - (i == j ? 1.0 : 0.0);
            ok &= fabs(r) <= 1e-12;
        }
    double cc[3], ch[3], cm[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double lo = crop.v[2 * r], hi = crop.v[2 * r + 1];
        ok &= (fabs(lo) < 1e300) & (fabs(hi) < 1e300);
        const double mid = 0.5 * (lo + hi);
        cc[r] = mid - t[r];
        ch[r] = fmax(0.5 * (hi - lo), 0.0);
        cm[r] = fabs(mid) + fabs(t[r]) + ch[r];
    }
    const double inf = __builtin_huge_val();
    double box[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double wc = inv[i][0] * cc[0] + inv[i][1] * cc[1] + inv[i][2] * cc[2];
        const double wh = fabs(inv[i][0]) * ch[0] + fabs(inv[i][1]) * ch[1] + fabs(inv[i][2]) * ch[2];
        const double mag = fabs(inv[i][0]) * cm[0] + fabs(inv[i][1]) * cm[1] + fabs(inv[i][2]) * cm[2];
        const double margin = 1e-6 + 1e-9 * mag;
        box[2 * i] = wc - wh - margin;
        box[2 * i + 1] = wc + wh + margin;
        ok &= (fabs(box[2 * i]) < 1e300) & (fabs(box[2 * i + 1]) < 1e300);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        frame_box[(size_t)f * 6 + 2 * i] = ok ? box[2 * i] : -inf;
        frame_box[(size_t)f * 6 + 2 * i + 1] = ok ? box[2 * i + 1] : inf;
    }
}

// Candidate pre-pass of site-sized maps (CAMA_BIN_WORKLIST).  k_block_cameras<true> below appends its survivors with one
// global atomic per (wave, work list), and returning atomics on one address retire at ~1 per 18 ns on this part: 23 k
// surviving (block, frame) items of a 4e6-vertex site map x 40 frames = 66 us of atomics around 5 us of arithmetic (the same
// kernel with the tests compiled out: 5.3 us), ~107 us beside an overlay.  So the map is first cut down with a test that
// costs six comparisons per (box, frame) -- box vs the frame's WORLD-space crop AABB (k_camera_functionals) -- by
// workgroups that loop over the frames (one bit per frame and box), and reserve list space ONCE (8 atomics per workgroup,
// a few hundred per launch); the exact tests then run on full waves of candidates only (k_candidate_cameras).
// Candidate (block b, frame f) goes to list b % 8, like the work lists.
// grid (ceil(4 * vblocks / BLOCK), ceil(F / CAND_FRAMES)); thread = box, looping over the frames of its chunk
constexpr int CAND_FRAMES = 64;
__global__ __launch_bounds__(BLOCK) void k_block_candidates(const double *__restrict__ bounds, const double *__restrict__ frame_box,
                                                            uint32_t F, uint32_t vblocks, uint32_t nsub, uint32_t list_cap,
                                                            uint32_t *__restrict__ cand_count, uint32_t *__restrict__ cand)
{
    __shared__ double s_fb[CAND_FRAMES * 6];
    __shared__ uint32_t s_cnt[BLOCK / 4], s_at[BLOCK / 4];      // per vertex block of this workgroup ("column")
    static_assert(CAND_FRAMES <= 64, "one bit per frame of the chunk");
    const uint32_t sb = blockIdx.x * BLOCK + threadIdx.x, b = sb >> 2, lane = __lane_id();
    const uint32_t f0 = blockIdx.y * (uint32_t)CAND_FRAMES, nf = min((uint32_t)CAND_FRAMES, F - f0);
    if (sb == 0 && f0 == 0) cand_count[8] = 1u;     // "cam_mask is only valid for the listed blocks" (cama_bin_stats)
    for (uint32_t t = threadIdx.x; t < nf * 6u; t += BLOCK) s_fb[t] = frame_box[(size_t)f0 * 6 + t];
    double bx[6] = {0, 0, 0, 0, 0, 0};
    const bool real = sb < nsub;
    if (real) {
#pragma unroll
        for (int j = 0; j < 6; ++j) bx[j] = bounds[(size_t)sb * 6 + j];
    }
    __syncthreads();
    // bit k = this box may reach into the crop box of frame f0 + k
    uint64_t bits = 0;
#pragma unroll 4
    for (uint32_t k = 0; k < nf; ++k) {
        const double *fb = s_fb + k * 6;            // LDS broadcast
        // every comparison is false on NaN: a box with a NaN bound stays a candidate
        const bool in = !((bx[1] < fb[0]) | (bx[0] > fb[1]) | (bx[3] < fb[2]) | (bx[2] > fb[3]) | (bx[5] < fb[4]) |
                          (bx[4] > fb[5]));
        bits |= (uint64_t)in << k;
    }
    if (!real) bits = 0;
    // a block is a candidate if any of its four boxes (adjacent lanes) is; its first lane owns it
    uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
    lo |= (uint32_t)__shfl_xor((int)lo, 1, 64); hi |= (uint32_t)__shfl_xor((int)hi, 1, 64);
    lo |= (uint32_t)__shfl_xor((int)lo, 2, 64); hi |= (uint32_t)__shfl_xor((int)hi, 2, 64);
    bits = ((lane & 3u) == 0u && b < vblocks) ? ((uint64_t)hi << 32 | lo) : 0ull;
    const uint32_t col = threadIdx.x >> 2, n = (uint32_t)__popcll(bits);
    if ((lane & 3u) == 0u) s_cnt[col] = n;
    __syncthreads();
    // list l = the columns with col % 8 == l (the workgroup's first block is a multiple of 64): thread l reserves for them
    if (threadIdx.x < 8u) {
        uint32_t tot = 0;
        for (uint32_t c = threadIdx.x; c < BLOCK / 4; c += 8u) tot += s_cnt[c];
        uint32_t at = tot ? atomicAdd(&cand_count[threadIdx.x], tot) : 0u;
        for (uint32_t c = threadIdx.x; c < BLOCK / 4; c += 8u) {
            s_at[c] = at;
            at += s_cnt[c];
        }
    }
    __syncthreads();
    if (!n) return;
    uint32_t *dst = cand + (size_t)(col & 7u) * list_cap + s_at[col];
    while (bits) {
        const uint32_t k = (uint32_t)__builtin_ctzll(bits);
        bits &= bits - 1ull;
        *dst++ = (f0 + k) * vblocks + b;
    }
}

// grid (ceil(4 * vblocks / BLOCK), F): one thread per (box, frame); the four boxes of a vertex block sit in adjacent lanes.
template <bool WORKLIST>
__global__ __launch_bounds__(BLOCK) void k_block_cameras(const double *__restrict__ bounds, const double *__restrict__ w2c,
                                                         const double *fn, int C, Crop crop, uint32_t vblocks, uint32_t nsub,
                                                         uint16_t *__restrict__ cam_mask, uint32_t list_cap,
                                                         uint32_t *__restrict__ work_count, uint32_t *__restrict__ work)
{
    const uint32_t sb = blockIdx.x * BLOCK + threadIdx.x, f = blockIdx.y;     // box; vertex block b = sb / 4
    const uint32_t b = sb >> 2;
    uint32_t mask = 0;
    if (sb < nsub) {
        const double *m = w2c + (size_t)f * 16, *bx = bounds + (size_t)sb * 6;
        const double wx = 0.5 * (bx[0] + bx[1]), wy = 0.5 * (bx[2] + bx[3]), wz = 0.5 * (bx[4] + bx[5]);
        const double ex = 0.5 * (bx[1] - bx[0]), ey = 0.5 * (bx[3] - bx[2]), ez = 0.5 * (bx[5] - bx[4]);
        mask = box_cameras(m, wx, wy, wz, ex, ey, ez, crop, fn, C);
    }
    // [F, vblocks] x 4 x u16 = the u64 per block the projection reads (boxes past the last vertex: 0)
    if (b < vblocks) cam_mask[((size_t)f * vblocks + b) * 4 + (sb & 3u)] = (uint16_t)mask;
    if (!WORKLIST) return;
    // a block goes on if any of its four boxes does; its first lane (lane % 4 == 0) appends it
    uint32_t any = mask | (uint32_t)__shfl_xor((int)mask, 1, 64);
    any |= (uint32_t)__shfl_xor((int)any, 2, 64);
    const uint32_t lane = __lane_id();
    const bool keep = any != 0u && (lane & 3u) == 0u && b < vblocks;
    const uint64_t mk = __ballot(keep);
    if (!mk) return;
    // one atomic per (wave, list): a wave holds 16 consecutive blocks, b % 8 == (lane / 4) % 8, so list l's members are
    // lanes 4 l and 4 l + 32; lane l < 8 reserves for list l
    const uint64_t pair = 0x0000000100000001ull;
    const uint64_t res = mk & (pair << (4u * (lane & 7u)));
    uint32_t base = 0;
    if (lane < 8u && res) base = atomicAdd(&work_count[lane], (uint32_t)__popcll(res));
    const uint32_t l = (lane >> 2) & 7u;
    base = __shfl(base, (int)l, 64);
    if (keep) {
        const uint64_t mine = mk & (pair << (4u * l));
        work[(size_t)l * list_cap + base + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull))] = f * vblocks + b;
    }
}

// The exact tests of k_block_cameras<true> on the candidate lists: workgroup g walks list g % 8 (gridDim.x is a multiple of
// 8), 64 candidates per iteration, thread 4 i + j = box j of candidate i.  Frames differ between lanes, so the frame's
// matrix comes through vector loads; the camera functionals stay scalar.  Writes the camera masks of the candidates (the
// projection reads cam_mask only for listed blocks) and appends the blocks with any camera left to work list g % 8 with ONE
// atomic per workgroup and iteration.
__global__ __launch_bounds__(BLOCK) void k_candidate_cameras(const double *__restrict__ bounds, const double *__restrict__ w2c,
                                                             const double *fn, int C, Crop crop, uint32_t vblocks, uint32_t nsub,
                                                             const uint32_t *__restrict__ cand_count,
                                                             const uint32_t *__restrict__ cand, uint16_t *__restrict__ cam_mask,
                                                             uint32_t list_cap, uint32_t *__restrict__ work_count,
                                                             uint32_t *__restrict__ work, uint32_t *__restrict__ frame_items,
                                                             uint32_t *__restrict__ work_rank,
                                                             unsigned long long *__restrict__ demand)
{
    __shared__ uint32_t s_cnt[2][BLOCK / 64], s_base[2], s_wc[2][BLOCK / 64];
    const uint32_t l = blockIdx.x & 7u, lane = __lane_id(), wave = threadIdx.x >> 6;
    const uint32_t n = ((const uint32_t __attribute__((address_space(4))) *)cand_count)[l];
    cand += (size_t)l * list_cap;
    work += (size_t)l * list_cap;
    work_rank += (size_t)l * list_cap;
    uint32_t par = 0;
    for (uint32_t i0 = (blockIdx.x >> 3) * (BLOCK / 4); i0 < n; i0 += (gridDim.x >> 3) * (BLOCK / 4), par ^= 1u) {
        const uint32_t i = i0 + (threadIdx.x >> 2);
        const bool live = i < n;
        const uint32_t item = live ? cand[i] : 0u;
        const uint32_t f = item / vblocks, b = item - f * vblocks, sb = 4u * b + (lane & 3u);
        uint32_t mask = 0;
        if (live && sb < nsub) {
            const double *m = w2c + (size_t)f * 16, *bx = bounds + (size_t)sb * 6;
            const double wx = 0.5 * (bx[0] + bx[1]), wy = 0.5 * (bx[2] + bx[3]), wz = 0.5 * (bx[4] + bx[5]);
            const double ex = 0.5 * (bx[1] - bx[0]), ey = 0.5 * (bx[3] - bx[2]), ez = 0.5 * (bx[5] - bx[4]);
            mask = box_cameras(m, wx, wy, wz, ex, ey, ez, crop, fn, C);
        }
        if (live) cam_mask[(size_t)item * 4 + (lane & 3u)] = (uint16_t)mask;
        uint32_t any = mask | (uint32_t)__shfl_xor((int)mask, 1, 64);
        any |= (uint32_t)__shfl_xor((int)any, 2, 64);
        const bool keep = any != 0u && (lane & 3u) == 0u;
        const uint64_t kk = __ballot(keep);
        // what the launch will need (the host sizes the stamp scratch from it before the projection runs): the (wave, camera)
        // chains of the listed blocks -- every lane is one wave of its block -- and each listed block's rank in its frame
        uint32_t wc = any != 0u ? (uint32_t)__popc(mask) : 0u;
#pragma unroll
        for (int d = 32; d; d >>= 1) wc += (uint32_t)__shfl_xor((int)wc, d, 64);
        if (lane == 0) { s_cnt[par][wave] = (uint32_t)__popcll(kk); s_wc[par][wave] = wc; }
        const uint32_t rank = keep ? atomicAdd(&frame_items[f], 1u) : 0u;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0, chains = 0;
#pragma unroll
            for (int w = 0; w < BLOCK / 64; ++w) { tot += s_cnt[par][w]; chains += s_wc[par][w]; }
            s_base[par] = tot ? atomicAdd(&work_count[l], tot) : 0u;
            if (chains) atomicAdd(demand, (unsigned long long)chains);
        }
        __syncthreads();
        // (the buffers of this parity are next written two iterations on, behind the next iteration's barriers)
        uint32_t at = s_base[par];
        for (uint32_t w = 0; w < wave; ++w) at += s_cnt[par][w];
        if (keep) {
            const uint32_t pos = at + (uint32_t)__popcll(kk & ((1ull << lane) - 1ull));
            work[pos] = item;
            work_rank[pos] = rank;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(BLOCK) void k_frames_project_list(FrameArgs a, const uint32_t *__restrict__ work_count,
                                                               const uint32_t *__restrict__ work, uint32_t vblocks,
                                                               uint32_t list_cap, const uint32_t *__restrict__ work_rank)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];
    const uint32_t list = blockIdx.x & 7u;          // gridDim.x is a multiple of 8
    const uint32_t n = work_count[list];
    work += (size_t)list * list_cap;
    if (work_rank) work_rank += (size_t)list * list_cap;
    for (uint32_t w = blockIdx.x >> 3; w < n; w += gridDim.x >> 3) {
        const uint32_t item = work[w];
        const int64_t vb = (int64_t)(item % vblocks);
        const int f = (int)(item / vblocks);
        // planned launches (work_rank): the block's segments are numbered by its rank among the frame's surviving blocks, so
        // a (frame, camera) row of the segment tables is as long as the busiest frame needs, not as long as the map
        const uint32_t seg_base = (work_rank ? work_rank[w] : (uint32_t)vb) * (BLOCK / SEG);
        hist_clear(a, s_hist);
        project_block<T>(a, vb, f, a.cam_mask[(size_t)f * a.vblocks + (size_t)vb], s_hist, seg_base);
        hist_flush(a, f, s_hist);
        __syncthreads();                            // the next item's clear must not overtake this flush
    }
}

// exclusive scan of each (frame,camera)'s band counters; one wave per (frame,camera)
__global__ __launch_bounds__(64) void k_scan_bands(const uint32_t *__restrict__ counts,
                                                   uint32_t *__restrict__ bin_off,
                                                   uint32_t *__restrict__ fc_total, int NB)
{
    const int fc = blockIdx.x, lane = threadIdx.x;
    uint32_t carry = 0;
    for (int base = 0; base < NB; base += 64) {
        const int b = base + lane;
        const uint32_t v = b < NB ? counts[(size_t)fc * NB + b] : 0u;
        uint32_t s = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(s, d, 64);
            if (lane >= d) s += t;
        }
        if (b < NB) bin_off[(size_t)fc * NB + b] = carry + s - v;
        carry += __shfl(s, 63, 64);
    }
    if (lane == 0) fc_total[fc] = carry;
}

// exclusive scan over the (frame,camera) totals; a single wave
__global__ __launch_bounds__(64) void k_scan_totals(const uint32_t *__restrict__ fc_total,
                                                    uint32_t *__restrict__ fc_base, int n)
{
    const int lane = threadIdx.x;
    uint32_t carry = 0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane;
        const uint32_t v = j < n ? fc_total[j] : 0u;
        uint32_t s = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(s, d, 64);
            if (lane >= d) s += t;
        }
        if (j < n) fc_base[j] = carry + s - v;
        carry += __shfl(s, 63, 64);
    }
}
