// raw35_kernels.hpp -- part of libcama_hip.so (included by cama_hip.hip inside its anonymous namespace).
// Raw-frame overlay for the reference's DEFAULT pipeline: 1600x900 sensor frames -> 960x540 tiles (scale 3/5, zero
// distortion; cama/reproject.py:164,176-182,232-240).  At that scale cv2.remap's fixed-point taps are a rational phase
// pattern: destination pixels 3k, 3k+1, 3k+2 read source pixels 5k | 5k+1,5k+2 | 5k+3,5k+4 with left/right weights
// 32/0 | 11/21 | 21/11 (out of 32), and the same holds for rows.  So 12 destination pixels (36 bytes = 9 dwords) come
// from exactly 20 source pixels (60 bytes = 15 dwords) of one or two source rows, nothing is shared between such units,
// and every tap position is a COMPILE-TIME byte offset: no per-pixel tap table and no per-tap addressing
// (k_overlay_rawlds spent ~50 LDS reads per 16 output bytes and was LDS-conflict bound, profiles/r02_before_raw_*).
// One workgroup = one band of R destination rows of one camera; the source rows the band needs are ONE contiguous byte
// range of the raw frame (bands do not share source rows), streamed into LDS with coalesced 16-byte loads.  One
// thread = one 12-pixel unit of one destination row: it reads its 2 x 15 dwords from LDS at dword stride 15 between
// lanes -- odd, hence bank-conflict free -- blends with v_dot4_u32_u8 against constant weight masks + 24-bit mads, and
// the 36 output bytes are transposed through the (by then dead) staging area into 16-byte streaming stores.
// (First cut read the 60-byte units straight from global memory: 16-byte loads at a 60-byte lane stride cost one L1
// request per LANE instead of per 4 lanes and ran at 0.41 ms, slower than the kernel it was to replace.)
// The host verifies the horizontal pattern (cama_raw35_plan) and supplies the per-row vertical taps and the per-band
// source rows; anything else takes the general kernels.  Arithmetic: remap_device.hpp (S = wt*t0 + wb*t1,
// t = wl*p0 + wr*p1, v = (S+512)>>10).
#pragma once

#ifndef RAW35_MAX_BLOCK
#define RAW35_MAX_BLOCK 640
#endif
#ifndef RAW35_STAGE_UNROLL
#define RAW35_STAGE_UNROLL 7      // 16-byte source chunks in flight per thread (7 rows x 300 chunks over 320 threads)
#endif

// channel `ch` of destination pixel k of a unit, horizontal part: t = wl * s[e] + wr * s[e + 3], e = 3 * q + ch
template <int K, int CH>
__device__ __forceinline__ uint32_t raw35_tap(const uint32_t (&d)[15])
{
    constexpr int g = K / 3, p = K % 3;
    constexpr int q = 5 * g + (p == 0 ? 0 : p == 1 ? 1 : 3);
    constexpr uint32_t wl = p == 0 ? 32u : p == 1 ? 11u : 21u, wr = 32u - wl;
    constexpr int e = 3 * q + CH, el = e >> 2, er = (e + 3) >> 2;
    constexpr uint32_t ml = wl << (8 * (e & 3)), mr = wr << (8 * ((e + 3) & 3));
    if constexpr (wr == 0u) {
        return __builtin_amdgcn_udot4(d[el], ml, 0u, false);
    } else if constexpr (el == er) {
        return __builtin_amdgcn_udot4(d[el], ml | mr, 0u, false);
    } else {
        return __builtin_amdgcn_udot4(d[er < 15 ? er : 14], mr, __builtin_amdgcn_udot4(d[el], ml, 0u, false), false);
    }
}

template <int K>
__device__ __forceinline__ uint32_t raw35_pixel(const uint32_t (&d0)[15], const uint32_t (&d1)[15], uint32_t wt, uint32_t wb)
{
    // weights <= 32, t <= 32*255: 24-bit multiply-adds; the rounding constant rides in the first mad
    const uint32_t vB = (__umul24(wb, raw35_tap<K, 0>(d1)) + (__umul24(wt, raw35_tap<K, 0>(d0)) + 512u)) >> 10;
    const uint32_t vG = (__umul24(wb, raw35_tap<K, 1>(d1)) + (__umul24(wt, raw35_tap<K, 1>(d0)) + 512u)) >> 10;
    const uint32_t vR = (__umul24(wb, raw35_tap<K, 2>(d1)) + (__umul24(wt, raw35_tap<K, 2>(d0)) + 512u)) >> 10;
    return vB | (vG << 8) | (vR << 16);
}

// four packed pixels (b | g<<8 | r<<16) -> the 12 bytes they occupy, as three dwords
__device__ __forceinline__ void raw35_pack4(const uint32_t *c, uint32_t *o)
{
    o[0] = c[0] | (c[1] << 24);
    o[1] = (c[1] >> 8) | (c[2] << 16);
    o[2] = (c[2] >> 16) | (c[3] << 8);
}

template <int... K>
__device__ __forceinline__ void raw35_unit(const uint32_t (&d0)[15], const uint32_t (&d1)[15], uint32_t wt, uint32_t wb,
                                           uint32_t (&px)[12], std::integer_sequence<int, K...>)
{
    ((px[K] = raw35_pixel<K>(d0, d1, wt, wb)), ...);
}

// A stamped band's owner table lives INSIDE the staging area, right behind the band's output rows (owner_off dwords; the
// staging area is always big enough at a 3:5 scale: rows * 1.25 W >= R * 1.75 W dwords, host-checked): it is built after
// every unit has read its taps, so a stamped band pays two more barriers and its rasterisation no longer overlaps the
// source loads, but every workgroup needs 38 KB of LDS instead of 54 KB -- 4 instead of 3 per CU: 140.0 -> 145.5 k
// frames/s at N = 10^4, 117.7 -> 123.5 k at N = 10^5 (same box).
__global__ __launch_bounds__(RAW35_MAX_BLOCK) void k_overlay_raw35(OverlayArgs a, const uint2 *__restrict__ vrows,
                                                                   const int2 *__restrict__ band_rows, int upr,
                                                                   int max_src_rows, int owner_off)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    // same workgroup order as k_overlay: (frame, mosaic row of cameras, band, camera column)
    const uint32_t cols = (uint32_t)a.cols, NB = (uint32_t)a.NB, C = (uint32_t)a.C;
    const uint32_t camrows = (C + cols - 1) / cols;
    uint32_t t = blockIdx.x;
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
    const int W = a.W;
    const uint32_t n = a.counts[bin];
    const uint32_t row_dwords = (uint32_t)W * 3u / 4u;                 // W % 48 == 0 (host-checked): multiple of 36
    const uint32_t src_row_dwords = (uint32_t)a.W0 * 3u / 4u;          // W0 * 3 % 16 == 0 (host-checked)
    uint32_t *s_stage = s_dyn;                                         // [max_src_rows * src_row_dwords], later the output
    uint32_t *s_owner = s_dyn + owner_off;                             // [R * W], stamped bands only

    // the band's first stamp record before the source loads (VMEM returns in order; see k_overlay)
    const uint2 *st = a.stamps + (n ? (size_t)a.fc_base[fc] + a.bin_off[bin] : (size_t)0);
    const uint2 first = st[n ? min(threadIdx.x, n - 1u) : 0u];
    __builtin_amdgcn_sched_barrier(0);

    // stage the band's source rows: one contiguous range of the raw frame, same layout in LDS
    const int2 br = band_rows[c * NB + b];                             // {first source row, number of source rows}
    const u32x4 *g = reinterpret_cast<const u32x4 *>(a.src + ((size_t)fc * a.H0 + br.x) * (size_t)a.W0 * 3);
    const uint32_t nsrc = (uint32_t)br.y * (src_row_dwords >> 2);      // 16-byte chunks
    constexpr int U = RAW35_STAGE_UNROLL;
    u32x4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) v[j] = OVERLAY_LOAD(g + min(threadIdx.x + j * blockDim.x, nsrc - 1u));

    u32x4 *s16 = reinterpret_cast<u32x4 *>(s_stage);
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const uint32_t idx = threadIdx.x + j * blockDim.x;
        if (idx < nsrc) s16[idx] = v[j];
    }
    for (uint32_t idx = threadIdx.x + U * blockDim.x; idx < nsrc; idx += blockDim.x) s16[idx] = OVERLAY_LOAD(g + idx);
    __syncthreads();

    // one 12-pixel unit per thread (the block covers the band: items <= blockDim, host-checked)
    const uint32_t items = (uint32_t)nrows * (uint32_t)upr;
    const bool active = threadIdx.x < items;
    const uint32_t item = active ? threadIdx.x : items - 1u;
    const uint32_t row = item / (uint32_t)upr, u = item - row * (uint32_t)upr;
    const uint2 vr = vrows[(size_t)c * a.H + y0 + (int)row];           // {r0 | r1 << 16, wt | wb << 8}
    const uint32_t wt = vr.y & 0xffu, wb = (vr.y >> 8) & 0xffu;
    // lane stride 15 dwords: odd, so the 32 banks are hit once each per half wave
    const uint32_t *p0 = s_stage + ((vr.x & 0xffffu) - (uint32_t)br.x) * src_row_dwords + u * 15u;
    const uint32_t *p1 = s_stage + ((vr.x >> 16) - (uint32_t)br.x) * src_row_dwords + u * 15u;
    uint32_t d0[15], d1[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) d0[k] = p0[k];
#pragma unroll
    for (int k = 0; k < 15; ++k) d1[k] = p1[k];
    if (n) {                                                           // (workgroup-uniform)
        __syncthreads();                                               // every unit holds its taps: staging is dead
        uint4 *o4 = reinterpret_cast<uint4 *>(s_owner);
        const int n4 = (nrows * W + 3) >> 2;
        for (int j = threadIdx.x; j < n4; j += blockDim.x) o4[j] = make_uint4(0, 0, 0, 0);
        lds_barrier();
        if (threadIdx.x < n) rasterise_one(s_owner, first, y0, nrows, W, a.disc);
        rasterise_rest(s_owner, st, threadIdx.x + blockDim.x, blockDim.x, n, y0, nrows, W, a.disc);
        __syncthreads();
    }
    uint32_t px[12];
    raw35_unit(d0, d1, wt, wb, px, std::make_integer_sequence<int, 12>{});
    if (n) {
        const uint4 *orow = reinterpret_cast<const uint4 *>(s_owner + row * (uint32_t)W + u * 12u);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint4 o = orow[j];
            if (o.x) px[4 * j + 0] = ((o.x - 1u) & 1u) ? a.pal.c[1] : a.pal.c[0];
            if (o.y) px[4 * j + 1] = ((o.y - 1u) & 1u) ? a.pal.c[1] : a.pal.c[0];
            if (o.z) px[4 * j + 2] = ((o.z - 1u) & 1u) ? a.pal.c[1] : a.pal.c[0];
            if (o.w) px[4 * j + 3] = ((o.w - 1u) & 1u) ? a.pal.c[1] : a.pal.c[0];
        }
    }
    uint32_t o[9];
    raw35_pack4(px, o);
    raw35_pack4(px + 4, o + 3);
    raw35_pack4(px + 8, o + 6);
    __syncthreads();                                                   // every unit has read its taps: staging is dead
    if (active) {
        uint32_t *dst = s_stage + row * row_dwords + u * 9u;           // R * row_dwords <= staging size (host-checked)
#pragma unroll
        for (int k = 0; k < 9; ++k) dst[k] = o[k];
    }
    __syncthreads();
    uint8_t *dcell = a.mosaic + (size_t)f * a.mosaic_frame_bytes +
                     ((size_t)(c / cols) * a.H + y0) * a.mosaic_row_bytes + (size_t)(c % cols) * W * 3;
    const uint32_t cpr = row_dwords >> 2;                               // 16-byte chunks per destination row
    const uint32_t nchunks = (uint32_t)nrows * cpr;
    uint32_t r = threadIdx.x / cpr, col = threadIdx.x - r * cpr;        // one division, then (row, chunk) advance by blockDim
    const uint32_t dr = blockDim.x / cpr, dc = blockDim.x - dr * cpr;
    for (uint32_t idx = threadIdx.x; idx < nchunks; idx += blockDim.x) {
        const u32x4 w = reinterpret_cast<const u32x4 *>(s_stage + r * row_dwords)[col];
        OVERLAY_STORE(w, reinterpret_cast<u32x4 *>(dcell + (size_t)r * a.mosaic_row_bytes) + col);
        r += dr;
        col += dc;
        if (col >= cpr) { col -= cpr; ++r; }
    }
}
