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

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "cama_hip.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail(CAMA_EHIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Opt-in timing of the dominant kernel (k_overlay) with HIP events recorded on the launch stream.
struct ProfileState {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;   // recorded, not yet collected
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
struct Disc { int radius; int hw[CAMA_MAX_RADIUS + 1]; };
struct Palette { uint32_t c[2]; };  // b | g<<8 | r<<16

// ------------------------------------------------------------------------------------------
// fp64 k-ordered FMA chains (see header: arithmetic contract)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void affine3x4(const double *M, double x, double y, double z,
                                          double &ox, double &oy, double &oz)
{
    double a;
    a = M[0] * x; a = __builtin_fma(M[1], y, a); a = __builtin_fma(M[2], z, a);  a = __builtin_fma(M[3], 1.0, a);  ox = a;
    a = M[4] * x; a = __builtin_fma(M[5], y, a); a = __builtin_fma(M[6], z, a);  a = __builtin_fma(M[7], 1.0, a);  oy = a;
    a = M[8] * x; a = __builtin_fma(M[9], y, a); a = __builtin_fma(M[10], z, a); a = __builtin_fma(M[11], 1.0, a); oz = a;
}

__device__ __forceinline__ void linear3x3(const double *K, double x, double y, double z,
                                          double &o0, double &o1, double &o2)
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
__device__ __forceinline__ bool visible_pixel(const double *m, double cx, double cy, double cz, double Wd, double Hd,
                                              uint32_t &uv)
{
    const double *K = m + 12;
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
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_transform_points(const T *__restrict__ xyz, int64_t N,
                                                            const double *__restrict__ Tm, Crop crop,
                                                            int has_crop, double *__restrict__ out,
                                                            uint8_t *__restrict__ mask)
{
    __shared__ double s_m[12];
    const int f = blockIdx.y;
    if (threadIdx.x < 12) s_m[threadIdx.x] = Tm[(size_t)f * 16 + threadIdx.x];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= N) return;
    const double x = (double)xyz[3 * i], y = (double)xyz[3 * i + 1], z = (double)xyz[3 * i + 2];
    double ox, oy, oz;
    affine3x4(s_m, x, y, z, ox, oy, oz);
    if (out) {
        double *o = out + ((size_t)f * N + i) * 3;
        o[0] = ox; o[1] = oy; o[2] = oz;
    }
    if (mask) mask[(size_t)f * N + i] = has_crop ? (uint8_t)in_crop(crop, ox, oy, oz) : (uint8_t)1;
}

__global__ __launch_bounds__(BLOCK) void k_crop_points(const double *__restrict__ pts, int64_t n, Crop crop,
                                                       uint8_t *__restrict__ mask)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) mask[i] = (uint8_t)in_crop(crop, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
}

__global__ __launch_bounds__(BLOCK) void k_project_points(const double *__restrict__ pts, int64_t n,
                                                          const double *__restrict__ c2cam,
                                                          const double *__restrict__ K, int C, int W, int H,
                                                          double *__restrict__ vu, uint8_t *__restrict__ vis)
{
    __shared__ double s_cam[CAMA_MAX_CAMERAS * CAM_STRIDE];
    stage_cameras(s_cam, c2cam, K, C);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
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
enum { MODE_EMIT = 0, MODE_COUNT = 1, MODE_FILL = 2 };

struct FrameArgs {
    const void *x, *y, *z;  // [N] each, float or double (template parameter T)
    const uint8_t *colour;
    const uint32_t *key;    // optional [N]: draw index << 1 | colour (maps stored in a different order than drawn)
    int64_t N;
    const double *w2c, *c2cam, *K;
    int C, W, H;
    Crop crop;
    // MODE_EMIT
    double *vu;
    uint8_t *vis, *crop_mask;
    // MODE_COUNT / MODE_FILL
    int band_shift, NB, radius;
    uint32_t *counts, *cursor;
    const uint32_t *bin_off, *fc_base;
    uint2 *stamps;
};

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

// Bin mode.  Every visible (vertex, camera) pair is a "stamp" {u:16, v:16, key = draw index << 1 | colour}
// that must reach the 1-2 row bands its disc touches.  Two passes (count -> scan -> fill) with identical
// arithmetic.  All per-stamp atomics are LDS atomics on a per-workgroup histogram of this frame's
// C x NB bins (ds_add_rtn gives the rank inside the workgroup); global memory sees one independent
// atomic per non-empty bin per workgroup, so nothing serialises on HBM/L2 latency.
constexpr int CAM_GROUP = 8;  // cameras ranked per pass through the workgroup protocol (register-resident entries)

template <int MODE, typename T>
__global__ __launch_bounds__(BLOCK) void k_frames_bin(FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];  // [C*NB] counts, then (fill) [C*NB] bases
    __shared__ double s_cam[CAMA_MAX_CAMERAS * CAM_STRIDE];
    const int f = blockIdx.y;
    const int nloc = a.C * a.NB;
    uint32_t *s_cnt = s_hist, *s_base = s_hist + nloc;

    // world->chassis + crop FIRST, with the frame's matrix read through wave-uniform (scalar) loads: on site-sized
    // maps ~95 % of the workgroups end here and never pay for staging the cameras or clearing the histogram
    const double *w2c = a.w2c + (size_t)f * 16;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    double cx = 0, cy = 0, cz = 0;
    bool in = false;
    uint32_t key = 0;
    if (i < a.N) {
        const double x = (double)static_cast<const T *>(a.x)[i], y = (double)static_cast<const T *>(a.y)[i],
                     z = (double)static_cast<const T *>(a.z)[i];
        affine3x4(w2c, x, y, z, cx, cy, cz);
        in = in_crop(a.crop, cx, cy, cz);
        // draw index << 1 | colour.  Spatially re-ordered maps carry it per vertex (a.key), otherwise it is the
        // storage index itself
        key = a.key ? a.key[i] : (((uint32_t)i << 1) | (uint32_t)(a.colour[i] & 1));
    }
    // whole workgroup outside the crop box (the common case on site-sized maps): done
    if (!__syncthreads_or((int)in)) return;
    stage_cameras(s_cam, a.c2cam, a.K, a.C);
    for (int t = threadIdx.x; t < nloc; t += BLOCK) s_cnt[t] = 0u;
    __syncthreads();

    const double Wd = (double)a.W, Hd = (double)a.H;
    const size_t gbin0 = (size_t)f * nloc;
    for (int c0 = 0; c0 < a.C; c0 += CAM_GROUP) {
        uint32_t e_uv[CAM_GROUP], e_slot[2 * CAM_GROUP];
#pragma unroll
        for (int j = 0; j < CAM_GROUP; ++j) {
            const int c = c0 + j;
            e_slot[2 * j] = e_slot[2 * j + 1] = 0xffffffffu;
            e_uv[j] = 0;
            // wave-uniform guard: the shuffle below must be executed by every lane
            if (c < a.C) {
                uint32_t uv = 0xffffffffu;      // packed truncated pixel, or "not visible"
                if (in) {
                    uint32_t packed;
                    if (visible_pixel(s_cam + c * CAM_STRIDE, cx, cy, cz, Wd, Hd, packed)) uv = packed;
                }
                // A disc is invisible if a LATER point (higher draw index) stamps the very same pixel (same
                // footprint).  The next lane is the next vertex of the polyline, so on dense maps (1 cm spacing) most
                // far-range stamps collapse here, exactly, before they cost atomics, HBM or LDS conflicts.
                const uint32_t uv_next = __shfl_down(uv, 1, 64);
                const uint32_t key_next = __shfl_down(key, 1, 64);
                const bool covered = (__lane_id() != 63u) && (uv_next == uv) && (key_next > key);
                const bool keep = (uv != 0xffffffffu) && !covered;
                if (keep) {
                    const int vi = (int)(uv >> 16);
                    const int b0 = max(vi - a.radius, 0) >> a.band_shift;
                    const int b1 = min(vi + a.radius, a.H - 1) >> a.band_shift;   // b1 <= b0 + 1 (host checks 2r <= R)
                    const uint32_t l0 = (uint32_t)(c * a.NB + b0);
                    if (MODE == MODE_COUNT) {
                        atomicAdd(&s_cnt[l0], 1u);
                        if (b1 != b0) atomicAdd(&s_cnt[l0 + 1], 1u);
                    } else {
                        e_uv[j] = uv;
                        e_slot[2 * j] = (l0 << 8) | atomicAdd(&s_cnt[l0], 1u);
                        if (b1 != b0) e_slot[2 * j + 1] = ((l0 + 1) << 8) | atomicAdd(&s_cnt[l0 + 1], 1u);
                    }
                }
            }
        }
        if (MODE == MODE_FILL) {
            __syncthreads();
            const int cend = min(c0 + CAM_GROUP, a.C);
            for (int t = c0 * a.NB + threadIdx.x; t < cend * a.NB; t += BLOCK) {
                const uint32_t n = s_cnt[t];
                uint32_t base = 0;
                if (n) base = atomicAdd(&a.cursor[gbin0 + t], n) + a.bin_off[gbin0 + t] + a.fc_base[f * a.C + t / a.NB];
                s_base[t] = base;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 2 * CAM_GROUP; ++j) {
                if (e_slot[j] != 0xffffffffu)
                    a.stamps[(size_t)s_base[e_slot[j] >> 8] + (e_slot[j] & 0xffu)] = make_uint2(e_uv[j >> 1], key);
            }
        }
    }
    if (MODE == MODE_COUNT) {
        __syncthreads();
        for (int t = threadIdx.x; t < nloc; t += BLOCK) {
            const uint32_t n = s_cnt[t];
            if (n) atomicAdd(&a.counts[gbin0 + t], n);
        }
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

// ------------------------------------------------------------------------------------------
// cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) building blocks (reproject.py:238-239), shared by the stand-alone
// resample kernel and by the overlay that reads raw sensor frames
// ------------------------------------------------------------------------------------------
struct Tap6 { uint32_t lo; uint32_t hi; };   // 6 useful bytes: pixel x0 (b,g,r) then pixel x0+1 (b,g,r)

// two horizontally adjacent BGR pixels starting at byte offset `o` of a frame of `frame_bytes` bytes (any alignment),
// always as two dword loads and without branches: a tap that would run past the end of the frame is read from
// `frame_bytes - 8` and funnel-shifted into place (only the last two pixels of a frame ever need it)
__device__ __forceinline__ Tap6 load_tap(const uint8_t *frame, size_t o, size_t frame_bytes)
{
    typedef uint32_t __attribute__((aligned(1))) u32u;
    const size_t last = frame_bytes - 8;
    const size_t at = o < last ? o : last;
    const uint32_t sh = (uint32_t)(o - at) * 8u;                      // 0, or 8..40 bits at the very end
    const uint64_t w = (uint64_t)*reinterpret_cast<const u32u *>(frame + at) |
                       ((uint64_t)*reinterpret_cast<const u32u *>(frame + at + 4) << 32);
    const uint64_t v = w >> sh;
    Tap6 t;
    t.lo = (uint32_t)v;
    t.hi = (uint32_t)(v >> 32);
    return t;
}

// Horizontal byte dot products + vertical blend of one pixel's 2x2 taps (see remap_pixel): t0 / t1 = six bytes
// [b0 g0 r0 b1 | g1 r1] of the top / bottom source row, wl/wr and wt/wb = tap weights out of 32 (0 when masked).
__device__ __forceinline__ uint32_t blend_taps(const Tap6 &t0, const Tap6 &t1, uint32_t wl, uint32_t wr, uint32_t wt,
                                               uint32_t wb)
{
    // weights on byte lanes: lo = [b0 g0 r0 b1], hi = [g1 r1 . .]
    const uint32_t wB = wl | (wr << 24), wGl = wl << 8, wGh = wr, wRl = wl << 16, wRh = wr << 8;
    const uint32_t tB0 = __builtin_amdgcn_udot4(t0.lo, wB, 0u, false);
    const uint32_t tG0 = __builtin_amdgcn_udot4(t0.hi, wGh, __builtin_amdgcn_udot4(t0.lo, wGl, 0u, false), false);
    const uint32_t tR0 = __builtin_amdgcn_udot4(t0.hi, wRh, __builtin_amdgcn_udot4(t0.lo, wRl, 0u, false), false);
    const uint32_t tB1 = __builtin_amdgcn_udot4(t1.lo, wB, 0u, false);
    const uint32_t tG1 = __builtin_amdgcn_udot4(t1.hi, wGh, __builtin_amdgcn_udot4(t1.lo, wGl, 0u, false), false);
    const uint32_t tR1 = __builtin_amdgcn_udot4(t1.hi, wRh, __builtin_amdgcn_udot4(t1.lo, wRl, 0u, false), false);
    // weights <= 32 and t <= 32*255: 24-bit multiply-adds (v_mad_u32_u24, full rate; v_mul_lo_u32 is quarter rate)
    const uint32_t vB = (__umul24(wt, tB0) + __umul24(wb, tB1) + 512u) >> 10;   // <= 255 by construction
    const uint32_t vG = (__umul24(wt, tG0) + __umul24(wb, tG1) + 512u) >> 10;
    const uint32_t vR = (__umul24(wt, tR0) + __umul24(wb, tR1) + 512u) >> 10;
    return vB | (vG << 8) | (vR << 16);
}

// One destination pixel of cv2.remap's 8-bit INTER_LINEAR path: returns b | g<<8 | r<<16.
// OpenCV: v = (w00 p00 + w01 p01 + w10 p10 + w11 p11 + 2^14) >> 15 with w00 = (32-a)(32-b)*32 etc.  That sum is
// exactly 32*S with  t_r = (32-a) p_r0 + a p_r1 (per source row r),  S = (32-b) t_0 + b t_1,  so v = (S + 512) >> 10.
// The horizontal step is a byte dot product: the six tap bytes [b0 g0 r0 b1 | g1 r1] against weights placed on the
// matching byte lanes (v_dot4_u32_u8), no unpacking.  Taps in the constant border contribute 0 = weight 0.
__device__ __forceinline__ uint32_t remap_pixel(const uint8_t *__restrict__ s, size_t frame_bytes, int H0, int W0,
                                                float mx, float my)
{
    const int sx = __float2int_rn(mx * 32.0f), sy = __float2int_rn(my * 32.0f);   // cvRound: half to even
    const int x0 = sx >> 5, y0 = sy >> 5;
    const bool xin0 = (unsigned)x0 < (unsigned)W0, xin1 = (unsigned)(x0 + 1) < (unsigned)W0;
    const bool yin0 = (unsigned)y0 < (unsigned)H0, yin1 = (unsigned)(y0 + 1) < (unsigned)H0;
    // (all four taps in the constant border: every weight below is 0 and the result is 0 -- no branch needed)
    const uint32_t a = (uint32_t)(sx & 31), b = (uint32_t)(sy & 31);
    uint32_t wl = xin0 ? 32u - a : 0u, wr = xin1 ? a : 0u;              // left / right tap weights
    const uint32_t wt = yin0 ? 32u - b : 0u, wb = yin1 ? b : 0u;        // top / bottom row weights
    // clamp the addresses into the frame (masked taps have weight 0, whatever bytes are read)
    const int xc = min(max(x0, 0), W0 - 1), y0c = min(max(y0, 0), H0 - 1), y1c = min(max(y0 + 1, 0), H0 - 1);
    const size_t o0 = ((size_t)y0c * W0 + xc) * 3, o1 = ((size_t)y1c * W0 + xc) * 3;
    const Tap6 t0 = load_tap(s, o0, frame_bytes), t1 = load_tap(s, o1, frame_bytes);
    const bool left_border = x0 < 0;      // x0 == -1: the in-range (right) tap is the FIRST pixel that was loaded
    wl = left_border ? wr : wl;
    wr = left_border ? 0u : wr;
    return blend_taps(t0, t1, wl, wr, wt, wb);
}

// Horizontal part of the remap of one destination COLUMN, packed: bits 0-15 source byte offset of the left tap
// (clamped into the row), 16-21 left weight, 22-27 right weight (out of 32; 0 for taps in the constant border; for
// x0 == -1 the in-range tap is the first loaded pixel, so the weights are swapped).  Depends only on mapx[x].
__device__ __forceinline__ uint32_t pack_column(float mx, int W0)
{
    const int sx = __float2int_rn(mx * 32.0f);                          // cvRound: half to even
    const int x0 = sx >> 5;
    const uint32_t a = (uint32_t)(sx & 31);
    const bool xin0 = (unsigned)x0 < (unsigned)W0, xin1 = (unsigned)(x0 + 1) < (unsigned)W0;
    uint32_t wl = xin0 ? 32u - a : 0u, wr = xin1 ? a : 0u;
    const bool left_border = x0 < 0;
    wl = left_border ? wr : wl;
    wr = left_border ? 0u : wr;
    const uint32_t off = (uint32_t)min(max(x0, 0), W0 - 1) * 3u;
    return off | (wl << 16) | (wr << 22);
}

// one destination pixel from source rows staged in LDS: row0 / row1 = the two staged rows (clamped into the staged
// range; wt / wb are 0 when the row is outside the frame), col = pack_column() of the destination column.
// 6 tap bytes per row at an arbitrary byte offset = three ALIGNED dword reads + funnel shifts (misaligned DS reads are
// split by the hardware and were 3-4x slower); rows start 16-byte aligned and the buffer is padded.
__device__ __forceinline__ uint32_t remap_pixel_lds(const uint8_t *row0, const uint8_t *row1, uint32_t col, uint32_t wt,
                                                    uint32_t wb)
{
    const uint32_t off = col & 0xffffu, sh = off & 3u, wl = (col >> 16) & 63u, wr = col >> 22;
    const uint32_t *q0 = reinterpret_cast<const uint32_t *>(row0 + (off & ~3u));
    const uint32_t *q1 = reinterpret_cast<const uint32_t *>(row1 + (off & ~3u));
    const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2], b0 = q1[0], b1 = q1[1], b2 = q1[2];
    Tap6 t0, t1;
    t0.lo = __builtin_amdgcn_alignbyte(a1, a0, sh); t0.hi = __builtin_amdgcn_alignbyte(a2, a1, sh);
    t1.lo = __builtin_amdgcn_alignbyte(b1, b0, sh); t1.hi = __builtin_amdgcn_alignbyte(b2, b1, sh);
    return blend_taps(t0, t1, wl, wr, wt, wb);
}


// map addressing: value for destination (y, x) is map[y * row_stride + x * col_stride]; full 2-D maps use (W, 1),
// separable ones (zero distortion: mapx = f(x), mapy = g(y)) use (0, 1) and (1, 0) on W- and H-long vectors
struct MapStride { int xr, xc, yr, yc; uint32_t w_magic; };

// ------------------------------------------------------------------------------------------
// overlay: band copy + deterministic stamp resolution
// ------------------------------------------------------------------------------------------
struct OverlayArgs {
    const uint8_t *src;
    uint8_t *mosaic;
    int C, H, W, cols, R, NB;
    uint32_t cpr, cpr_magic;          // 16-byte chunks per row, ceil(2^32 / cpr)
    size_t mosaic_row_bytes, mosaic_frame_bytes;
    const uint32_t *counts, *bin_off, *fc_base;
    const uint2 *stamps;
    Disc disc;
    Palette pal;
    // RESAMPLE variant: src holds RAW frames [F,C,H0,W0,3]; each mosaic pixel is remapped from them on the fly
    int H0, W0;
    const float *mapx, *mapy;            // per camera: mapx + c * mapx_cam, mapy + c * mapy_cam
    int64_t mapx_cam, mapy_cam;
    MapStride ms;
};

template <int THREADS = OVERLAY_BLOCK>
__device__ __forceinline__ void rasterise_stamps(uint32_t *s_owner, const uint2 *st, uint32_t n,
                                                 int y0, int nrows, int W, const Disc &disc)
{
    for (uint32_t s = threadIdx.x; s < n; s += THREADS) {
        const uint2 r = st[s];
        const int u = (int)(r.x & 0xffffu), v = (int)(r.x >> 16);
        const uint32_t val = r.y + 1u;  // 0 = no owner
        const int ylo = max(v - disc.radius, y0), yhi = min(v + disc.radius, y0 + nrows - 1);
        for (int y = ylo; y <= yhi; ++y) {
            const int hw = disc.hw[abs(y - v)];
            if (hw < 0) continue;
            const int xlo = max(u - hw, 0), xhi = min(u + hw, W - 1);
            uint32_t *row = s_owner + (y - y0) * W;
            for (int x = xlo; x <= xhi; ++x) atomicMax(&row[x], val);
        }
    }
}

// Patch the 16 bytes of chunk `col` of one row with the colours of the owned pixels it overlaps.
// A chunk starts at byte 16*col = 3*p0 + ph and overlaps exactly pixels p0..p0+5.
__device__ __forceinline__ void patch_chunk(u32x4 &d, const uint32_t *orow, uint32_t col, const Palette &pal)
{
    const uint32_t b0 = col * 16u, p0 = b0 / 3u, ph = b0 - p0 * 3u;
    uint32_t o[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = orow[p0 + k];
    if ((o[0] | o[1] | o[2] | o[3] | o[4] | o[5]) == 0u) return;
    uint32_t c[6], m[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        m[k] = o[k] ? 0x00ffffffu : 0u;
        c[k] = o[k] ? pal.c[(o[k] - 1u) & 1u] : 0u;
    }
    // 18-byte little-endian streams (pixel k at bytes 3k..3k+2) as 5 dwords
    const uint32_t V0 = c[0] | (c[1] << 24), V1 = (c[1] >> 8) | (c[2] << 16), V2 = (c[2] >> 16) | (c[3] << 8),
                   V3 = c[4] | (c[5] << 24), V4 = c[5] >> 8;
    const uint32_t M0 = m[0] | (m[1] << 24), M1 = (m[1] >> 8) | (m[2] << 16), M2 = (m[2] >> 16) | (m[3] << 8),
                   M3 = m[4] | (m[5] << 24), M4 = m[5] >> 8;
    // chunk byte j is stream byte j + ph: funnel-shift right by ph bytes
    const uint32_t v0 = __builtin_amdgcn_alignbyte(V1, V0, ph), v1 = __builtin_amdgcn_alignbyte(V2, V1, ph),
                   v2 = __builtin_amdgcn_alignbyte(V3, V2, ph), v3 = __builtin_amdgcn_alignbyte(V4, V3, ph);
    const uint32_t m0 = __builtin_amdgcn_alignbyte(M1, M0, ph), m1 = __builtin_amdgcn_alignbyte(M2, M1, ph),
                   m2 = __builtin_amdgcn_alignbyte(M3, M2, ph), m3 = __builtin_amdgcn_alignbyte(M4, M3, ph);
    d.x = (d.x & ~m0) | (v0 & m0);
    d.y = (d.y & ~m1) | (v1 & m1);
    d.z = (d.z & ~m2) | (v2 & m2);
    d.w = (d.w & ~m3) | (v3 & m3);
}

// 6 packed pixels (b | g<<8 | r<<16) that a 16-byte chunk starting `ph` bytes into the first one overlaps -> the chunk
__device__ __forceinline__ u32x4 chunk_from_pixels(const uint32_t *c, uint32_t ph)
{
    const uint32_t V0 = c[0] | (c[1] << 24), V1 = (c[1] >> 8) | (c[2] << 16), V2 = (c[2] >> 16) | (c[3] << 8),
                   V3 = c[4] | (c[5] << 24), V4 = c[5] >> 8;
    u32x4 v;
    v.x = __builtin_amdgcn_alignbyte(V1, V0, ph);
    v.y = __builtin_amdgcn_alignbyte(V2, V1, ph);
    v.z = __builtin_amdgcn_alignbyte(V3, V2, ph);
    v.w = __builtin_amdgcn_alignbyte(V4, V3, ph);
    return v;
}

template <bool VEC, bool RESAMPLE>
__global__ __launch_bounds__(OVERLAY_BLOCK) void k_overlay(OverlayArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_owner[];  // R x W, used only by stamped bands
#ifdef OVERLAY_ORDER_FCB
    const uint32_t bin = blockIdx.x;
    const uint32_t fc = bin / (uint32_t)a.NB, b = bin - fc * (uint32_t)a.NB;
    const uint32_t f = fc / (uint32_t)a.C, c = fc - f * (uint32_t)a.C;
#else
    // Workgroup order (frame, mosaic row of cameras, band, camera column): the `cols` cameras that share a mosaic
    // row-band are adjacent in launch order, so R full mosaic rows (R * cols*W*3 contiguous bytes) are written
    // close together in time instead of one third at a time.
    const uint32_t cols = (uint32_t)a.cols, NB = (uint32_t)a.NB, C = (uint32_t)a.C;
    const uint32_t camrows = (C + cols - 1) / cols;
    uint32_t t = blockIdx.x;
    const uint32_t cc = t % cols; t /= cols;
    const uint32_t b = t % NB;    t /= NB;
    const uint32_t cr = t % camrows;
    const uint32_t f = t / camrows;
    const uint32_t c = cr * cols + cc;
    if (c >= C) return;                                  // ragged last camera row
    const uint32_t fc = f * C + c;
    const uint32_t bin = fc * NB + b;
#endif
    const int y0 = (int)b * a.R;
    const int nrows = min(a.R, a.H - y0);
    const int W = a.W;
    const uint32_t n = a.counts[bin];

    if (n) {
        uint4 *o4 = reinterpret_cast<uint4 *>(s_owner);
        const int n4 = (nrows * W + 3) >> 2;
        for (int j = threadIdx.x; j < n4; j += OVERLAY_BLOCK) o4[j] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        rasterise_stamps(s_owner, a.stamps + ((size_t)a.fc_base[fc] + a.bin_off[bin]), n, y0, nrows, W, a.disc);
        __syncthreads();
    }

    const uint8_t *sband = a.src + ((size_t)fc * a.H + y0) * (size_t)W * 3;      // (unused by RESAMPLE)
    uint8_t *dcell = a.mosaic + (size_t)f * a.mosaic_frame_bytes +
                     ((size_t)(c / (uint32_t)a.cols) * a.H + y0) * a.mosaic_row_bytes +
                     (size_t)(c % (uint32_t)a.cols) * W * 3;

    if (RESAMPLE) {
        // The source is the raw sensor frame of (f, c): every destination pixel is the fixed-point bilinear blend of
        // its 2x2 source taps (cv2.remap semantics), computed here instead of being read from a pre-resized frame.
        // A 16-byte chunk overlaps 6 destination pixels; adjacent lanes own adjacent chunks, so their taps are
        // adjacent in the raw frame.  Raw bytes are read once from HBM (re-reads hit L1/L2), resized frames never
        // exist in memory.
        const size_t raw_frame = (size_t)a.H0 * a.W0 * 3;
        const uint8_t *raw = a.src + (size_t)fc * raw_frame;
        const float *mxc = a.mapx + (size_t)c * a.mapx_cam, *myc = a.mapy + (size_t)c * a.mapy_cam;
        const uint32_t nchunks = (uint32_t)nrows * a.cpr;
        for (uint32_t idx = threadIdx.x; idx < nchunks; idx += OVERLAY_BLOCK) {
            const uint32_t row = __umulhi(idx, a.cpr_magic), col = idx - row * a.cpr;
            const uint32_t b0 = col * 16u, p0 = b0 / 3u, ph = b0 - p0 * 3u;
            const int y = y0 + (int)row;
            float mx[6], my[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int x = (int)p0 + k;
                mx[k] = mxc[y * a.ms.xr + x * a.ms.xc];
                my[k] = myc[y * a.ms.yr + x * a.ms.yc];
            }
            uint32_t px[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) px[k] = remap_pixel(raw, raw_frame, a.H0, a.W0, mx[k], my[k]);
            u32x4 v = chunk_from_pixels(px, ph);
            if (n) patch_chunk(v, s_owner + row * W, col, a.pal);
            u32x4 *drow = reinterpret_cast<u32x4 *>(dcell + (size_t)row * a.mosaic_row_bytes);
            OVERLAY_STORE(v, drow + col);
        }
        return;
    }

    if (VEC) {
        // the band is one contiguous byte range in src: chunk j of the band is src16[j]
        constexpr int U = OVERLAY_UNROLL;
        const u32x4 *s16 = reinterpret_cast<const u32x4 *>(sband);
        const uint32_t nchunks = (uint32_t)nrows * a.cpr;
        for (uint32_t base = threadIdx.x; base < nchunks; base += OVERLAY_BLOCK * U) {
            u32x4 v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const uint32_t idx = base + j * OVERLAY_BLOCK;
                if (idx < nchunks) v[j] = OVERLAY_LOAD(s16 + idx);
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const uint32_t idx = base + j * OVERLAY_BLOCK;
                if (idx < nchunks) {
                    const uint32_t row = __umulhi(idx, a.cpr_magic);
                    const uint32_t col = idx - row * a.cpr;
                    if (n) patch_chunk(v[j], s_owner + row * W, col, a.pal);
                    u32x4 *drow = reinterpret_cast<u32x4 *>(dcell + (size_t)row * a.mosaic_row_bytes);
                    OVERLAY_STORE(v[j], drow + col);
                }
            }
        }
    } else {
        // generic width / alignment: one pixel per thread-iteration
        const int npix = nrows * W;
        for (int p = threadIdx.x; p < npix; p += OVERLAY_BLOCK) {
            const int row = p / W, x = p - row * W;
            const uint8_t *s = sband + (size_t)p * 3;
            uint8_t b0 = s[0], b1 = s[1], b2 = s[2];
            if (n) {
                const uint32_t o = s_owner[p];
                if (o) {
                    const uint32_t col = a.pal.c[(o - 1u) & 1u];
                    b0 = (uint8_t)col; b1 = (uint8_t)(col >> 8); b2 = (uint8_t)(col >> 16);
                }
            }
            uint8_t *d = dcell + (size_t)row * a.mosaic_row_bytes + (size_t)x * 3;
            d[0] = b0; d[1] = b1; d[2] = b2;
        }
    }
}

// Raw-frame overlay, LDS-staged (separable maps = zero lens distortion, the nuScenes / CAMA calibration):
// the source rows a band of R destination rows needs (host-precomputed [first, count] per camera and band) are
// streamed into LDS once with 16-byte loads -- the only global reads of image data -- and every bilinear tap is an
// LDS read.  The gather variant (k_overlay<true,true>) spends ~40 VMEM instructions per 16 output bytes; this one
// spends one per 16 INPUT bytes.  LDS: owner table R*W*4 + staged rows + the camera's mapx vector.
#ifndef RAWLDS_BLOCK
#define RAWLDS_BLOCK 256
#endif
// Raw-frame overlay, LDS-staged (separable maps = zero lens distortion, the nuScenes / CAMA calibration).
// One workgroup per (frame, camera, band of R destination rows, column tile of Wt destination columns): the source
// rows x source byte range that tile needs (host-precomputed per camera/band and per camera/tile) are streamed into
// LDS once with 16-byte loads -- the only global reads of image data -- and every bilinear tap is an aligned LDS dword
// read.  The gather variant (k_overlay<true,true>) spends ~40 VMEM instructions per 16 output bytes; this one spends
// one per 16 INPUT bytes.  Column tiles keep a workgroup's LDS near 20 KB (7 workgroups per CU) instead of 62 KB.
// LDS: owner table R*Wt*4 | per-column packed taps Wt*4 | staged source rows.
__global__ __launch_bounds__(RAWLDS_BLOCK) void k_overlay_rawlds(OverlayArgs a, const int2 *__restrict__ band_rows,
                                                                 const int2 *__restrict__ tile_bytes, int TX, int Wt)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const uint32_t cols = (uint32_t)a.cols, NB = (uint32_t)a.NB, C = (uint32_t)a.C;
    const uint32_t camrows = (C + cols - 1) / cols;
    uint32_t t = blockIdx.x;
    const uint32_t tx = t % (uint32_t)TX; t /= (uint32_t)TX;
    const uint32_t cc = t % cols; t /= cols;
    const uint32_t b = t % NB;    t /= NB;
    const uint32_t cr = t % camrows;
    const uint32_t f = t / camrows;
    const uint32_t c = cr * cols + cc;
    if (c >= C) return;
    const uint32_t fc = f * C + c;
    const uint32_t bin = fc * NB + b;
    const int y0 = (int)b * a.R;
    const int nrows = min(a.R, a.H - y0);
    const int W0 = a.W0, x_first = (int)tx * Wt;
    const uint32_t n = a.counts[bin];
    const size_t row_bytes = (size_t)W0 * 3;

    const int2 br = band_rows[c * NB + b];          // first source row, number of source rows
    const int2 tb = tile_bytes[c * TX + tx];        // first source byte within a row (16-aligned), bytes (x16)
    const uint32_t stride = (uint32_t)tb.y;         // LDS row stride

    uint32_t *s_owner = s_dyn;                                               // [R*Wt]
    uint32_t *s_col = s_dyn + (((size_t)a.R * Wt + 3) & ~(size_t)3);        // [Wt] pack_column(), offsets tile-relative
    uint8_t *s_src = static_cast<uint8_t *>(__builtin_assume_aligned(        // [rows * stride + 16]
        reinterpret_cast<uint8_t *>(s_col + ((Wt + 3) & ~3)), 16));

    // stage the tile's source bytes row by row + this tile's column taps
    const uint8_t *g0 = a.src + (size_t)fc * a.H0 * row_bytes + (size_t)br.x * row_bytes + (size_t)tb.x;
    const uint32_t cpr_src = stride >> 4, nchunk_src = (uint32_t)br.y * cpr_src;
    for (uint32_t i = threadIdx.x; i < nchunk_src; i += RAWLDS_BLOCK) {
        const uint32_t r = i / cpr_src, j = i - r * cpr_src;
        reinterpret_cast<u32x4 *>(s_src + (size_t)r * stride)[j] =
            OVERLAY_LOAD(reinterpret_cast<const u32x4 *>(g0 + (size_t)r * row_bytes) + j);
    }
    const float *mxc = a.mapx + (size_t)c * a.mapx_cam, *myc = a.mapy + (size_t)c * a.mapy_cam;
    for (int x = threadIdx.x; x < Wt; x += RAWLDS_BLOCK) s_col[x] = pack_column(mxc[x_first + x], W0) - (uint32_t)tb.x;
    if (n) {
        uint4 *o4 = reinterpret_cast<uint4 *>(s_owner);
        const int n4 = (nrows * Wt + 3) >> 2;
        for (int j = threadIdx.x; j < n4; j += RAWLDS_BLOCK) o4[j] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        // the band's stamps, clipped to this tile's columns
        const uint2 *st = a.stamps + ((size_t)a.fc_base[fc] + a.bin_off[bin]);
        for (uint32_t s = threadIdx.x; s < n; s += RAWLDS_BLOCK) {
            const uint2 rec = st[s];
            const int u = (int)(rec.x & 0xffffu), v = (int)(rec.x >> 16);
            const uint32_t val = rec.y + 1u;
            const int ylo = max(v - a.disc.radius, y0), yhi = min(v + a.disc.radius, y0 + nrows - 1);
            for (int y = ylo; y <= yhi; ++y) {
                const int hw = a.disc.hw[abs(y - v)];
                if (hw < 0) continue;
                const int xlo = max(max(u - hw, 0), x_first), xhi = min(min(u + hw, a.W - 1), x_first + Wt - 1);
                uint32_t *orow = s_owner + (y - y0) * Wt - x_first;
                for (int x = xlo; x <= xhi; ++x) atomicMax(&orow[x], val);
            }
        }
    }
    __syncthreads();

    uint8_t *dcell = a.mosaic + (size_t)f * a.mosaic_frame_bytes +
                     ((size_t)(c / cols) * a.H + y0) * a.mosaic_row_bytes + (size_t)(c % cols) * a.W * 3 +
                     (size_t)x_first * 3;
    const uint32_t cpr_t = (uint32_t)(Wt * 3) >> 4;             // 16-byte chunks per tile row
    const uint32_t nchunks = (uint32_t)nrows * cpr_t;
    const int ylast = br.x + br.y - 1;
    for (uint32_t idx = threadIdx.x; idx < nchunks; idx += RAWLDS_BLOCK) {
        const uint32_t row = idx / cpr_t, col = idx - row * cpr_t;
        const uint32_t b0 = col * 16u, p0 = b0 / 3u, ph = b0 - p0 * 3u;
        // vertical part of the remap: once per chunk (all six pixels share the destination row)
        const int sy = __float2int_rn(myc[y0 + (int)row] * 32.0f);
        const int yy0 = sy >> 5;
        const uint32_t bw = (uint32_t)(sy & 31);
        const uint32_t wt = ((unsigned)yy0 < (unsigned)a.H0) ? 32u - bw : 0u;
        const uint32_t wb = ((unsigned)(yy0 + 1) < (unsigned)a.H0) ? bw : 0u;
        const uint8_t *row0 = s_src + __umul24((uint32_t)(min(max(yy0, br.x), ylast) - br.x), stride);
        const uint8_t *row1 = s_src + __umul24((uint32_t)(min(max(yy0 + 1, br.x), ylast) - br.x), stride);
        uint32_t px[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) px[k] = remap_pixel_lds(row0, row1, s_col[p0 + k], wt, wb);
        u32x4 v = chunk_from_pixels(px, ph);
        if (n) patch_chunk(v, s_owner + row * Wt, col, a.pal);
        u32x4 *drow = reinterpret_cast<u32x4 *>(dcell + (size_t)row * a.mosaic_row_bytes);
        OVERLAY_STORE(v, drow + col);
    }
}

// ------------------------------------------------------------------------------------------
// generic single-image stamping (CameraManager.render_maps on caller-supplied points)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_stamp_global(const double *__restrict__ vu,
                                                        const uint8_t *__restrict__ colour, int64_t n,
                                                        uint32_t *__restrict__ owner, int H, int W, Disc disc)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    // reproject.py:249 astype(np.int32); cv2.circle clips to the image itself
    const int v = (int)vu[2 * i], u = (int)vu[2 * i + 1];
    const uint32_t val = ((((uint32_t)i) << 1) | (uint32_t)(colour[i] & 1)) + 1u;
    for (int dy = -disc.radius; dy <= disc.radius; ++dy) {
        const int y = v + dy;
        if (y < 0 || y >= H) continue;
        const int hw = disc.hw[abs(dy)];
        if (hw < 0) continue;
        const int xlo = max(u - hw, 0), xhi = min(u + hw, W - 1);
        for (int x = xlo; x <= xhi; ++x) atomicMax(&owner[(size_t)y * W + x], val);
    }
}

__global__ __launch_bounds__(BLOCK) void k_apply_owner(const uint32_t *__restrict__ owner,
                                                       uint8_t *__restrict__ image, int64_t npix, Palette pal)
{
    const int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p >= npix) return;
    const uint32_t o = owner[p];
    if (!o) return;
    const uint32_t col = pal.c[(o - 1u) & 1u];
    image[3 * p] = (uint8_t)col;
    image[3 * p + 1] = (uint8_t)(col >> 8);
    image[3 * p + 2] = (uint8_t)(col >> 16);
}

// ------------------------------------------------------------------------------------------
// frame resample: cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) over float32 maps (reproject.py:238-239)
// OpenCV's 8-bit remap quantises coordinates to 1/32 px (INTER_BITS = 5) and blends with 15-bit fixed-point
// weights; for bilinear the weights (32-a)(32-b)*32 ... are exact integers summing to 1 << 15.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_resample(const uint8_t *__restrict__ src, int64_t src_stride,
                                                    uint8_t *__restrict__ dst, int64_t dst_stride, int H0, int W0,
                                                    int H, int W, const float *__restrict__ mapx,
                                                    const float *__restrict__ mapy, MapStride ms)
{
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= H * W) return;
    const uint8_t *s = src + (size_t)blockIdx.y * src_stride;
    uint8_t *d = dst + (size_t)blockIdx.y * dst_stride + (size_t)p * 3;
    const int y = (int)__umulhi((uint32_t)p, ms.w_magic), x = p - y * W;
    const uint32_t c = remap_pixel(s, (size_t)H0 * W0 * 3, H0, W0, mapx[y * ms.xr + x * ms.xc],
                                   mapy[y * ms.yr + x * ms.yc]);
    d[0] = (uint8_t)c; d[1] = (uint8_t)(c >> 8); d[2] = (uint8_t)(c >> 16);
}

// W % 16 == 0: thread <-> destination pixel for the gathers (adjacent lanes read adjacent source pixels, the map
// loads are coalesced), PPT pixels per thread strided by the workgroup size so the dependent load chains
// (map -> taps) of several pixels overlap; then the workgroup's bytes are transposed through LDS into aligned
// 16-byte stores.  (One thread per 16 CONSECUTIVE pixels was tried: lanes 80 source bytes apart lose all
// coalescing -- 4.7x slower; one pixel per thread is latency-bound at 8 workgroups/CU.)
#ifndef RESAMPLE_PPT_N
#define RESAMPLE_PPT_N 4
#endif
constexpr int RESAMPLE_PPT = RESAMPLE_PPT_N;

__global__ __launch_bounds__(BLOCK) void k_resample16(const uint8_t *__restrict__ src, int64_t src_stride,
                                                      uint8_t *__restrict__ dst, int64_t dst_stride, int H0, int W0,
                                                      int H, int W, const float *__restrict__ mapx,
                                                      const float *__restrict__ mapy, MapStride ms)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_out[BLOCK * RESAMPLE_PPT * 3];
    const int npix = H * W;
    const int first = blockIdx.x * (BLOCK * RESAMPLE_PPT);
    const uint8_t *s = src + (size_t)blockIdx.y * src_stride;
    float mx[RESAMPLE_PPT], my[RESAMPLE_PPT];
#pragma unroll
    for (int j = 0; j < RESAMPLE_PPT; ++j) {
        const int p = min(first + j * BLOCK + (int)threadIdx.x, npix - 1);
        const int y = (int)__umulhi((uint32_t)p, ms.w_magic), x = p - y * W;
        mx[j] = mapx[y * ms.xr + x * ms.xc];
        my[j] = mapy[y * ms.yr + x * ms.yc];
    }
#pragma unroll
    for (int j = 0; j < RESAMPLE_PPT; ++j) {
        const uint32_t c = remap_pixel(s, (size_t)H0 * W0 * 3, H0, W0, mx[j], my[j]);
        uint8_t *o = s_out + 3 * (j * BLOCK + (int)threadIdx.x);
        o[0] = (uint8_t)c; o[1] = (uint8_t)(c >> 8); o[2] = (uint8_t)(c >> 16);
    }
    __syncthreads();
    // npix is a multiple of 16, so the tail workgroup ends on a chunk boundary
    const int valid_chunks = (min(BLOCK * RESAMPLE_PPT, npix - first) * 3) >> 4;
    u32x4 *d = reinterpret_cast<u32x4 *>(dst + (size_t)blockIdx.y * dst_stride + (size_t)first * 3);
    for (int k = threadIdx.x; k < valid_chunks; k += BLOCK) d[k] = reinterpret_cast<const u32x4 *>(s_out)[k];
}

// ------------------------------------------------------------------------------------------
// static-map build (per clip): densify labels, lift with the BEV height raster, pixel -> world
// reproject.py:42-106.  One thread per OUTPUT point; float32 arithmetic in the reference's operation order
// (compiled -ffp-contract=off; HIP's float division is correctly rounded), so the buffer is bit-identical to the
// host build.  Writes the SoA vertex buffer + colour ids the fused render consumes: the map never visits the host.
// ------------------------------------------------------------------------------------------
struct MapBuildArgs {
    const float *verts;        // [V,2] label vertices (float32, as np.array(data).astype(np.float32))
    const int32_t *seg_v0;     // [S] first vertex of each non-empty segment (its end is v0 + 1)
    const int32_t *seg_num;    // [S] points emitted by the segment = int(|seg| / solution) > 0
    const int64_t *seg_off;    // [S+1] exclusive scan of seg_num
    const uint8_t *seg_colour; // [S]
    int32_t S;
    int64_t N;
    int32_t lift;              // 1: CAMA labels (BEV pixels + raster), 0: nuScenes labels (metres, z = 0)
    const void *raster;        // [rows, cols] float32 / float64
    int32_t rows, cols;
    float solution, half_w, half_h, cx, cy;
    void *x, *y, *z;           // [N] each, float32 or float64 (TZ)
    uint8_t *colour;           // [N]
};

template <typename TZ>
__global__ __launch_bounds__(BLOCK) void k_build_map(MapBuildArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.N) return;
    // segment that owns output point i: last s with seg_off[s] <= i
    int lo = 0, hi = a.S - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.seg_off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    const int s = lo;
    const float j = (float)(i - a.seg_off[s]);
    const float num = (float)a.seg_num[s];
    const float2 p0 = reinterpret_cast<const float2 *>(a.verts)[a.seg_v0[s]];
    const float2 p1 = reinterpret_cast<const float2 *>(a.verts)[a.seg_v0[s] + 1];
    // start + (end - start) / num * j      (reproject.py:62 / :92)
    const float px = p0.x + ((p1.x - p0.x) / num) * j;
    const float py = p0.y + ((p1.y - p0.y) / num) * j;
    TZ ox, oy, oz;
    if (a.lift) {
        // round().astype(np.uint16)[:, ::-1].clip(0, rows-1): half-to-even, C cast through int32 (wraps), (row, col)
        const int row = min(max((int)(uint16_t)(int32_t)rintf(py), 0), a.rows - 1);
        const int col = min(max((int)(uint16_t)(int32_t)rintf(px), 0), a.rows - 1);
        oz = static_cast<const TZ *>(a.raster)[(size_t)row * a.cols + col];
        // world x from pixel y and vice versa (reproject.py:38-39), float32
        ox = (TZ)(((py * a.solution) - a.half_w) + a.cx);
        oy = (TZ)(((px * a.solution) - a.half_h) + a.cy);
    } else {
        ox = (TZ)px; oy = (TZ)py; oz = (TZ)0;
    }
    static_cast<TZ *>(a.x)[i] = ox;
    static_cast<TZ *>(a.y)[i] = oy;
    static_cast<TZ *>(a.z)[i] = oz;
    a.colour[i] = a.seg_colour[s];
}

// ------------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------------
int make_disc(int radius, const int32_t *hw, Disc &d)
{
    if (radius < 0 || radius > CAMA_MAX_RADIUS || !hw) return -1;
    d.radius = radius;
    for (int k = 0; k <= CAMA_MAX_RADIUS; ++k) d.hw[k] = (k <= radius) ? hw[k] : -1;
    return 0;
}

Palette make_palette(const uint8_t *bgr)
{
    Palette p;
    for (int k = 0; k < 2; ++k)
        p.c[k] = (uint32_t)bgr[3 * k] | ((uint32_t)bgr[3 * k + 1] << 8) | ((uint32_t)bgr[3 * k + 2] << 16);
    return p;
}

int band_rows_for(int W)
{
    const char *env = getenv("CAMA_BAND_ROWS");
    if (env) {
        int r = atoi(env);
        if (r == 4 || r == 8 || r == 16 || r == 32) return r;
    }
    // Measured on MI355X (profiles/): ~20 KB of image per workgroup streams best (more, smaller
    // workgroups balance stamped bands and keep the LDS owner table R*W*4 <= 26 KB -> 6 workgroups per CU).
    // R must stay >= 2*radius so a disc touches at most two bands.
    if (W >= 600) return 4;
    if (W >= 300) return 8;
    return 16;
}

int log2i(int v)
{
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct ScratchLayout {
    size_t counts, cursor, bin_off, fc_total, fc_base, stamps, total;
    uint64_t capacity;
    int R, NB, bands_per_stamp;
};

int layout_scratch(int64_t N, int F, int C, int H, int W, int radius, ScratchLayout &L)
{
    L.R = band_rows_for(W);
    L.NB = (H + L.R - 1) / L.R;
    L.bands_per_stamp = radius > 0 ? 2 : 1;  // 2r+1 rows touch <= 2 bands when 2r <= R (checked by the caller)
    const size_t nbins = (size_t)F * C * L.NB, nfc = (size_t)F * C;
    L.capacity = (uint64_t)F * C * (uint64_t)N * L.bands_per_stamp;
    size_t off = 0;
    L.counts = off;   off = align_up(off + nbins * 4, 256);
    L.cursor = off;   off = align_up(off + nbins * 4, 256);
    L.bin_off = off;  off = align_up(off + nbins * 4, 256);
    L.fc_total = off; off = align_up(off + nfc * 4, 256);
    L.fc_base = off;  off = align_up(off + nfc * 4, 256);
    L.stamps = off;   off = align_up(off + (size_t)L.capacity * 8, 256);
    L.total = off;
    return 0;
}

int check_common(int64_t N, int F, int C, int W, int H)
{
    if (N < 0 || N >= (1ll << 30)) return fail(CAMA_EINVAL, "N=%lld out of range [0, 2^30)", (long long)N);
    if (F < 0 || F > 65535) return fail(CAMA_EINVAL, "F=%d out of range [0, 65535]", F);
    if (C < 1 || C > CAMA_MAX_CAMERAS) return fail(CAMA_EINVAL, "C=%d out of range [1, %d]", C, CAMA_MAX_CAMERAS);
    if (W < 1 || H < 1 || W > 65535 || H > 65535) return fail(CAMA_EINVAL, "W x H = %d x %d out of range", W, H);
    return 0;
}

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

int cama_profile_collect(double *total_ms, int32_t *launches)
{
    double sum = 0.0;
    int n = 0;
    for (auto &pr : g_prof.pending) {
        HIP_TRY(hipEventSynchronize(pr.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        sum += ms;
        ++n;
        g_prof.pool.push_back(pr.first);
        g_prof.pool.push_back(pr.second);
    }
    g_prof.pending.clear();
    if (total_ms) *total_ms = sum;
    if (launches) *launches = n;
    return CAMA_OK;
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
static int check_render(int64_t N, int32_t F, int32_t C, int32_t W, int32_t H, int32_t radius, const void *scratch,
                        size_t scratch_bytes, ScratchLayout &L)
{
    if (int rc = check_common(N, F, C, W, H)) return rc;
    if (radius < 0 || radius > CAMA_MAX_RADIUS) return fail(CAMA_EINVAL, "radius %d out of range", radius);
    if (!scratch) return fail(CAMA_EINVAL, "scratch is NULL");
    layout_scratch(N, F, C, H, W, radius, L);
    if (scratch_bytes < L.total) return fail(CAMA_EINVAL, "scratch too small: %zu < %zu", scratch_bytes, L.total);
    if (L.capacity >= (1ull << 32))
        return fail(CAMA_EINVAL, "F*C*N*%d = %llu stamps exceed 32-bit offsets: render fewer frames per call",
                    L.bands_per_stamp, (unsigned long long)L.capacity);
    if ((size_t)F * C * L.NB >= (1ull << 31)) return fail(CAMA_EINVAL, "too many bands");
    if (2 * radius > L.R)
        return fail(CAMA_EINVAL, "radius %d too large for the fused path (needs 2r <= band rows = %d)", radius, L.R);
    if ((size_t)C * L.NB * 8 > 64 * 1024)
        return fail(CAMA_EINVAL, "C*bands = %d*%d exceeds the per-workgroup LDS histogram", C, L.NB);
    if (align_up((size_t)L.R * W * 4, 16) > 160 * 1024) return fail(CAMA_EINVAL, "W=%d too wide for the LDS owner table", W);
    return CAMA_OK;
}

int cama_bin_frames(const void *x, const void *y, const void *z, int32_t xyz_is_f64, const uint8_t *colour_id,
                    const uint32_t *draw_key, int64_t N, const double *w2c, int32_t F, const double *c2cam, const double *K, int32_t C,
                    const double *crop, int32_t W, int32_t H, int32_t radius, void *scratch, size_t scratch_bytes,
                    void *stream)
{
    ScratchLayout L;
    if (int rc = check_render(N, F, C, W, H, radius, scratch, scratch_bytes, L)) return rc;
    if (F == 0) return CAMA_OK;
    if (!w2c || !c2cam || !K || !crop) return fail(CAMA_EINVAL, "NULL pointer argument");
    if (N && (!x || !y || !z || (!colour_id && !draw_key))) return fail(CAMA_EINVAL, "NULL vertex buffer");

    hipStream_t s = (hipStream_t)stream;
    char *base = (char *)scratch;
    uint32_t *counts = (uint32_t *)(base + L.counts), *cursor = (uint32_t *)(base + L.cursor);
    uint32_t *bin_off = (uint32_t *)(base + L.bin_off), *fc_total = (uint32_t *)(base + L.fc_total);
    uint32_t *fc_base = (uint32_t *)(base + L.fc_base);
    const int nfc = F * C;

    // counts and cursor are adjacent: one memset
    HIP_TRY(hipMemsetAsync(counts, 0, L.bin_off - L.counts, s));

    FrameArgs a{};
    a.x = x; a.y = y; a.z = z; a.colour = colour_id; a.key = draw_key; a.N = N;
    a.w2c = w2c; a.c2cam = c2cam; a.K = K; a.C = C; a.W = W; a.H = H;
    memcpy(a.crop.v, crop, sizeof(a.crop.v));
    a.band_shift = log2i(L.R); a.NB = L.NB; a.radius = radius;
    a.counts = counts; a.cursor = cursor; a.bin_off = bin_off; a.fc_base = fc_base;
    a.stamps = (uint2 *)(base + L.stamps);
    // XCD-aware mapping: workgroup (x, y) has linear id x + y * gridDim.x and runs on XCD id % 8.  With gridDim.x a
    // multiple of 8, vertex chunk x is processed on the SAME XCD for every frame y, so on big maps each XCD's 4 MB L2
    // keeps its 1/8 of the vertex buffer across all frames instead of re-fetching it per frame (padding workgroups
    // exit at once: no vertex, no survivor).
    const unsigned vblocks = (unsigned)((N + BLOCK - 1) / BLOCK);
    const dim3 fgrid(getenv("CAMA_NO_XCD_PAD") ? vblocks : ((vblocks + 7u) & ~7u), (unsigned)F);
    const size_t hist_lds = align_up((size_t)C * L.NB * 4, 16);
    if (N) {
        if (xyz_is_f64)
            hipLaunchKernelGGL((k_frames_bin<MODE_COUNT, double>), fgrid, dim3(BLOCK), hist_lds, s, a);
        else
            hipLaunchKernelGGL((k_frames_bin<MODE_COUNT, float>), fgrid, dim3(BLOCK), hist_lds, s, a);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_scan_bands, dim3(nfc), dim3(64), 0, s, counts, bin_off, fc_total, L.NB);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(64), 0, s, fc_total, fc_base, nfc);
    HIP_TRY(hipGetLastError());
    if (N) {
        if (xyz_is_f64)
            hipLaunchKernelGGL((k_frames_bin<MODE_FILL, double>), fgrid, dim3(BLOCK), 2 * hist_lds, s, a);
        else
            hipLaunchKernelGGL((k_frames_bin<MODE_FILL, float>), fgrid, dim3(BLOCK), 2 * hist_lds, s, a);
        HIP_TRY(hipGetLastError());
    }
    return CAMA_OK;
}

struct RawSource {
    int H0, W0;
    const float *mapx, *mapy;
    int separable;
    const int32_t *band_rows;   // [C,NB,2] or NULL
    int max_src_rows;
    const int32_t *tile_bytes;  // [C,TX,2] or NULL
    int tiles_x, max_tile_bytes;
};

static int overlay_impl(const uint8_t *src, const RawSource *raw, uint8_t *mosaic, int64_t N, int32_t F, int32_t C,
                        int32_t H, int32_t W, int32_t cols, int32_t radius, const int32_t *halfwidth,
                        const uint8_t *palette_bgr, const void *scratch, size_t scratch_bytes, void *stream)
{
    ScratchLayout L;
    if (int rc = check_render(N, F, C, W, H, radius, scratch, scratch_bytes, L)) return rc;
    if (F == 0) return CAMA_OK;
    if (cols < 1) return fail(CAMA_EINVAL, "cols=%d", cols);
    if (!src || !mosaic || !palette_bgr) return fail(CAMA_EINVAL, "NULL pointer argument");
    Disc disc;
    if (make_disc(radius, halfwidth, disc)) return fail(CAMA_EINVAL, "bad radius/halfwidth table");
    const size_t lds = align_up((size_t)L.R * W * 4, 16);
    hipStream_t s = (hipStream_t)stream;
    const char *base = (const char *)scratch;
#ifdef OVERLAY_ORDER_FCB
    const int nfc = F * C;
#endif

    OverlayArgs o{};
    o.src = src; o.mosaic = mosaic; o.C = C; o.H = H; o.W = W; o.cols = cols; o.R = L.R; o.NB = L.NB;
    const int rows = (C + cols - 1) / cols;
    o.mosaic_row_bytes = (size_t)cols * W * 3;
    o.mosaic_frame_bytes = (size_t)rows * H * o.mosaic_row_bytes;
    o.counts = (const uint32_t *)(base + L.counts); o.bin_off = (const uint32_t *)(base + L.bin_off);
    o.fc_base = (const uint32_t *)(base + L.fc_base); o.stamps = (const uint2 *)(base + L.stamps);
    o.disc = disc; o.pal = make_palette(palette_bgr);
    const bool vec = (W % 16 == 0) && ((((raw ? 0 : (uintptr_t)src)) | (uintptr_t)mosaic) % 16 == 0);
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
#ifdef OVERLAY_ORDER_FCB
    const unsigned nblocks = (unsigned)((size_t)nfc * L.NB);
#else
    const unsigned nblocks = (unsigned)((size_t)F * rows * cols * L.NB);   // (frame, camera row, band, camera column)
#endif
    if (lds > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_overlay<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void *)k_overlay<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void *)k_overlay<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (g_prof.on) {
        ev0 = prof_event();
        ev1 = prof_event();
        if (ev0 && ev1) HIP_TRY(hipEventRecord(ev0, s));
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
    if (raw_lds) {
        if (lds_raw > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)k_overlay_rawlds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds_raw));
        hipLaunchKernelGGL(k_overlay_rawlds, dim3(nblocks * (unsigned)raw->tiles_x), dim3(RAWLDS_BLOCK), lds_raw, s, o,
                           reinterpret_cast<const int2 *>(raw->band_rows),
                           reinterpret_cast<const int2 *>(raw->tile_bytes), raw->tiles_x, Wt);
    } else if (raw)
        hipLaunchKernelGGL((k_overlay<true, true>), dim3(nblocks), dim3(OVERLAY_BLOCK), lds, s, o);
    else if (vec)
        hipLaunchKernelGGL((k_overlay<true, false>), dim3(nblocks), dim3(OVERLAY_BLOCK), lds, s, o);
    else
        hipLaunchKernelGGL((k_overlay<false, false>), dim3(nblocks), dim3(OVERLAY_BLOCK), lds, s, o);
    HIP_TRY(hipGetLastError());
    if (ev0 && ev1) {
        HIP_TRY(hipEventRecord(ev1, s));
        g_prof.pending.emplace_back(ev0, ev1);
    }
    return CAMA_OK;
}

int cama_overlay_frames(const uint8_t *src, uint8_t *mosaic, int64_t N, int32_t F, int32_t C, int32_t H, int32_t W,
                        int32_t cols, int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr,
                        const void *scratch, size_t scratch_bytes, void *stream)
{
    return overlay_impl(src, nullptr, mosaic, N, F, C, H, W, cols, radius, halfwidth, palette_bgr, scratch,
                        scratch_bytes, stream);
}

int cama_overlay_frames_raw(const uint8_t *raw, int32_t H0, int32_t W0, const float *mapx, const float *mapy,
                            int32_t separable, const int32_t *band_src_rows, int32_t max_src_rows,
                            const int32_t *tile_src_bytes, int32_t tiles_x, int32_t max_tile_bytes, uint8_t *mosaic,
                            int64_t N, int32_t F, int32_t C, int32_t H, int32_t W, int32_t cols, int32_t radius,
                            const int32_t *halfwidth, const uint8_t *palette_bgr, const void *scratch,
                            size_t scratch_bytes, void *stream)
{
    const RawSource rs{H0, W0, mapx, mapy, separable, band_src_rows, max_src_rows, tile_src_bytes, tiles_x, max_tile_bytes};
    return overlay_impl(raw, &rs, mosaic, N, F, C, H, W, cols, radius, halfwidth, palette_bgr, scratch, scratch_bytes,
                        stream);
}

int cama_render_frames(const void *x, const void *y, const void *z, int32_t xyz_is_f64, const uint8_t *colour_id,
                       const uint32_t *draw_key, int64_t N, const double *w2c, int32_t F, const double *c2cam, const double *K, int32_t C,
                       const double *crop, int32_t W, int32_t H, const uint8_t *src, uint8_t *mosaic, int32_t cols,
                       int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr, void *scratch,
                       size_t scratch_bytes, void *stream)
{
    // validate the overlay half first so that nothing is enqueued when it would be rejected
    if (F > 0 && (!src || !mosaic || !palette_bgr || !halfwidth || cols < 1))
        return check_common(N, F, C, W, H) ? CAMA_EINVAL : fail(CAMA_EINVAL, "NULL pointer argument");
    if (int rc = cama_bin_frames(x, y, z, xyz_is_f64, colour_id, draw_key, N, w2c, F, c2cam, K, C, crop, W, H, radius, scratch,
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

size_t cama_stamp_scratch_bytes(int32_t H, int32_t W)
{
    if (H < 1 || W < 1) return 0;
    return align_up((size_t)H * W * 4, 256);
}

int cama_stamp_points(const double *vu, const uint8_t *colour_id, int64_t n, uint8_t *image, int32_t H, int32_t W,
                      int32_t radius, const int32_t *halfwidth, const uint8_t *palette_bgr, void *scratch,
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
    const int64_t npix = (int64_t)H * W;
    hipLaunchKernelGGL(k_apply_owner, dim3((unsigned)((npix + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s,
                       (const uint32_t *)scratch, image, npix, make_palette(palette_bgr));
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

}  // extern "C"
