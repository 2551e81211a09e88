// overlay_kernels.hpp -- part of libcama_hip.so (included by cama_hip.hip inside its anonymous namespace).
// Overlay kernels: band copy + deterministic stamp resolution (plain, raw-frame gather, raw-frame LDS-staged)
// and the generic single-image stamping.
#pragma once

// ------------------------------------------------------------------------------------------
// overlay: band copy + deterministic stamp resolution
// ------------------------------------------------------------------------------------------
struct OverlayArgs {
    int f0;                           // multi-scene chains: launch-wide number of this launch's first frame (the scratch is
                                      // indexed by the launch-wide frame, src / mosaic by the frame inside the scene)
    uint32_t chunk_log2;              // items per XCD chunk = 2^chunk_log2 (>= 31: one contiguous range per XCD)
    uint32_t items;                   // bands of this launch = F * camera rows * cols * NB (x column tiles); grid = 8 * ceil(items / 8)
    uint32_t cols_magic, nb_magic, cr_magic;     // ceil(2^32 / d) for d = cols, NB, camera rows (divmod_magic)
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

// one stamp's disc -> atomicMax(draw key) over its footprint rows inside the band
__device__ __forceinline__ void rasterise_one(uint32_t *s_owner, const uint2 r, int y0, int nrows, int W, const Disc &disc)
{
    const int u = (int)(r.x & 0xffffu), v = (int)(r.x >> 16);
    const uint32_t val = r.y + 1u;  // 0 = no owner
    const int ylo = max(v - disc.radius, y0), yhi = min(v + disc.radius, y0 + nrows - 1);
    for (int y = ylo; y <= yhi; ++y) {
        const int hw = disc_halfwidth(disc, abs(y - v));
        if (hw < 0) continue;
        const int xlo = max(u - hw, 0), xhi = min(u + hw, W - 1);
        uint32_t *row = s_owner + (y - y0) * W;
        // (tried: a plain read first and the atomic only where the key would grow -- owners are monotone, so that is
        // exact, and on dense maps ~9 of 10 atomics would go away -- but the dependent read costs more than the
        // fire-and-forget ds_max it saves: dense 10^6 overlay 441 -> 498 us)
        for (int x = xlo; x <= xhi; ++x) atomicMax(&row[x], val);
    }
}

// The same on a table whose rows carry `radius` spare cells on either side (row stride Wp = W + 2 * radius, pixel x at
// cell x + radius): spans never need clamping, and a row's span is the centre plus symmetric pairs.  The rasteriser is
// instruction-bound on maps that stack hundreds of thousands of stamps per frame (ablation on the dense 10^6 map: the
// stamp loop is ALL of the 108 us that separate its overlay from a stamp-free copy), and this form has about half the
// instructions per (stamp, row): no x clamps, no per-pixel loop counter, one table lookup from a 32-bit packed register.
// hw8 = half widths of rows |dy| = 0..7, 4 bits each (radius <= 7); rowmask bit k = row |dy| = k is drawn.
__device__ __forceinline__ void rasterise_one_padded(uint32_t *s_owner, const uint2 r, int y0, int nrows, int Wp, int radius,
                                                     uint32_t hw8, uint32_t rowmask)
{
    const int u = (int)(r.x & 0xffffu), v = (int)(r.x >> 16);
    const uint32_t val = r.y + 1u;
    const int ylo = max(v - radius, y0), yhi = min(v + radius, y0 + nrows - 1);
    uint32_t *cell = s_owner + (ylo - y0) * Wp + u + radius;           // centre column of row ylo
    for (int y = ylo; y <= yhi; ++y, cell += Wp) {
        const uint32_t k = (uint32_t)abs(y - v);
        if (!((rowmask >> k) & 1u)) continue;
        const int hw = (int)((hw8 >> (4u * k)) & 15u);
        atomicMax(cell, val);
        for (int d = 1; d <= hw; ++d) {
            atomicMax(cell - d, val);
            atomicMax(cell + d, val);
        }
    }
}

// The reference's disc (cv2.circle radius 2, cama/reproject.py:256: rows of half width 0 / 1 / 2 / 1 / 0, 13 pixels) with
// nothing left to loop over: five row tests, thirteen ds_max at immediate offsets from five row addresses.  The generic
// loops above compile to ~120 instructions per stamp (per row: |dy|, mask test, half-width extract, exec juggling; per pixel
// pair: two address adds, a loop counter, a branch); this is ~40 -- and the rasteriser is instruction-bound on stamp-heavy
// bands (site maps: 7 k stamps per image, most of them in a few horizon bands).  Same cells, same values: max is order-free.
__device__ __forceinline__ void rasterise_r2_padded(uint32_t *s_owner, const uint2 r, int y0, int nrows, int Wp)
{
    const int rel = (int)(r.x >> 16) - y0;                              // centre row relative to the band: -2 .. nrows + 1
    const int c = rel * Wp + (int)(r.x & 0xffffu) + 2;                  // centre cell (pixel x at cell x + radius)
    const uint32_t val = r.y + 1u;
    const unsigned n = (unsigned)nrows;
    if ((unsigned)rel < n) {
        atomicMax(&s_owner[c - 2], val); atomicMax(&s_owner[c - 1], val); atomicMax(&s_owner[c], val);
        atomicMax(&s_owner[c + 1], val); atomicMax(&s_owner[c + 2], val);
    }
    if ((unsigned)(rel - 1) < n) {
        const int p = c - Wp;
        atomicMax(&s_owner[p - 1], val); atomicMax(&s_owner[p], val); atomicMax(&s_owner[p + 1], val);
    }
    if ((unsigned)(rel + 1) < n) {
        const int p = c + Wp;
        atomicMax(&s_owner[p - 1], val); atomicMax(&s_owner[p], val); atomicMax(&s_owner[p + 1], val);
    }
    if ((unsigned)(rel - 2) < n) atomicMax(&s_owner[c - 2 * Wp], val);
    if ((unsigned)(rel + 2) < n) atomicMax(&s_owner[c + 2 * Wp], val);
}

// is this the reference's radius-2 disc?  (wave-uniform: kernel arguments)
__device__ __forceinline__ bool disc_is_r2(int radius, uint32_t hw8, uint32_t rowmask)
{
    return radius == 2 && (hw8 & 0xfffu) == 0x012u && (rowmask & 7u) == 7u;
}

// EXTENSION (segment records, 16 bytes): the one-pixel 8-connected Bresenham segment from the predecessor's pixel (r.z) to the
// point's own (r.x), both ends included, under the point's key -- the integer recurrence of oracle_line_bresenham, restated;
// only the rows of this band are written (the other bands the segment crosses hold a copy of the record and do theirs).
__device__ __forceinline__ void rasterise_segment_padded(uint32_t *s_owner, const uint4 r, int y0, int nrows, int Wp, int radius)
{
    if (r.z == 0xffffffffu) return;
    int x = (int)(r.z & 0xffffu), y = (int)(r.z >> 16);
    const int x1 = (int)(r.x & 0xffffu), y1 = (int)(r.x >> 16);
    const uint32_t val = r.y + 1u;
    const int dx = abs(x1 - x), sx = x < x1 ? 1 : -1;
    const int dy = -abs(y1 - y), sy = y < y1 ? 1 : -1;
    int err = dx + dy;
    for (int guard = 0; guard < 4 * 65536; ++guard) {                  // (a segment has at most W + H pixels)
        if ((unsigned)(y - y0) < (unsigned)nrows) atomicMax(&s_owner[(y - y0) * Wp + x + radius], val);
        else if ((sy > 0 && y >= y0 + nrows) || (sy < 0 && y < y0)) break;     // past the band, moving away from it
        if (x == x1 && y == y1) break;
        const int e2 = 2 * err;
        if (e2 >= dy) { err += dy; x += sx; }
        if (e2 <= dx) { err += dx; y += sy; }
    }
}

// EXTENSION, anti-aliased variant (Wu): the claims of oracle_render_frame_wu's wu_line_claims -- the same integer recurrence --
// for the rows of this band; a claim is ((key + 1) << 8) | coverage, zero coverages claim nothing, atomicMax keeps the
// greatest (key, coverage) per pixel.  Columns -1 and W fall into the owner rows' padding (radius >= 1 cells either side).
__device__ __forceinline__ void wu_claim_padded(uint32_t *s_owner, int x, int y, uint32_t v, int y0, int nrows, int Wp, int radius)
{
    if ((unsigned)(y - y0) < (unsigned)nrows && x >= -radius && x < Wp - radius && (v & 255u))
        atomicMax(&s_owner[(y - y0) * Wp + x + radius], v);
}

__device__ __forceinline__ void rasterise_segment_wu_padded(uint32_t *s_owner, const uint4 r, int y0, int nrows, int Wp, int radius)
{
    if (r.z == 0xffffffffu) return;
    const int x0 = (int)(r.z & 0xffffu), yy0 = (int)(r.z >> 16), x1 = (int)(r.x & 0xffffu), yy1 = (int)(r.x >> 16);
    const uint32_t key1 = (r.y + 1u) << 8;
    const bool steep = abs(yy1 - yy0) > abs(x1 - x0);
    int a0 = steep ? yy0 : x0, b0 = steep ? x0 : yy0, a1 = steep ? yy1 : x1, b1 = steep ? x1 : yy1;      // a = major, b = minor
    if (a0 > a1) { int t = a0; a0 = a1; a1 = t; t = b0; b0 = b1; b1 = t; }
    const int da = a1 - a0, db = b1 - b0;
    if (da == 0) return;                                     // (same pixel: the record carries no segment then)
    const int num = db * 65536;
    int grad = num / da;
    if (num % da != 0 && num < 0) grad -= 1;                 // floor division, da > 0
    // steep: the major axis is the row -- only the band's rows are walked
    const int j_lo = steep ? max(0, y0 - a0) : 0, j_hi = steep ? min(da, y0 + nrows - 1 - a0) : da;
    for (int j = j_lo; j <= j_hi; ++j) {
        const int y = b0 * 65536 + grad * j;
        const int row = y >> 16;                             // arithmetic shift: floor
        const uint32_t f = ((uint32_t)y & 0xffffu) >> 8;
        const int a = a0 + j;
        if (steep) {
            wu_claim_padded(s_owner, row, a, key1 | (255u - f), y0, nrows, Wp, radius);
            wu_claim_padded(s_owner, row + 1, a, key1 | f, y0, nrows, Wp, radius);
        } else {
            wu_claim_padded(s_owner, a, row, key1 | (255u - f), y0, nrows, Wp, radius);
            wu_claim_padded(s_owner, a, row + 1, key1 | f, y0, nrows, Wp, radius);
        }
    }
}

// the disc of a record in the anti-aliased variant: coverage 255 under ((key + 1) << 8)
__device__ __forceinline__ void rasterise_disc_wu_padded(uint32_t *s_owner, const uint2 r, int y0, int nrows, int Wp, int radius,
                                                         uint32_t hw8, uint32_t rowmask)
{
    const int u = (int)(r.x & 0xffffu), v = (int)(r.x >> 16);
    const uint32_t val = ((r.y + 1u) << 8) | 255u;
    const int ylo = max(v - radius, y0), yhi = min(v + radius, y0 + nrows - 1);
    uint32_t *cell = s_owner + (ylo - y0) * Wp + u + radius;
    for (int y = ylo; y <= yhi; ++y, cell += Wp) {
        const uint32_t k = (uint32_t)abs(y - v);
        if (!((rowmask >> k) & 1u)) continue;
        const int hw = (int)((hw8 >> (4u * k)) & 15u);
        for (int d = -hw; d <= hw; ++d) atomicMax(cell + d, val);
    }
}

// Stamps `first`, first + stride, ... < n of a band's list, four record loads in flight per thread: on bands that
// collect tens of thousands of stamps (dense maps: every far lane converges on a few horizon rows) the loop is a chain of
// global-load latencies, and such a band's workgroup is the kernel's straggler.  Loads are unconditional (clamped index):
// a load under a divergent guard is waited for at the end of the guard.
__device__ __forceinline__ void rasterise_rest_padded(uint32_t *s_owner, const uint2 *st, uint32_t first, uint32_t stride,
                                                      uint32_t n, int y0, int nrows, int Wp, int radius, uint32_t hw8,
                                                      uint32_t rowmask)
{
    if (disc_is_r2(radius, hw8, rowmask)) {                              // the reference's disc: the unrolled diamond
        for (uint32_t s = first; s < n; s += 4u * stride) {
            const uint2 r0 = st[s], r1 = st[min(s + stride, n - 1u)], r2 = st[min(s + 2u * stride, n - 1u)],
                        r3 = st[min(s + 3u * stride, n - 1u)];
            rasterise_r2_padded(s_owner, r0, y0, nrows, Wp);
            if (s + stride < n) rasterise_r2_padded(s_owner, r1, y0, nrows, Wp);
            if (s + 2u * stride < n) rasterise_r2_padded(s_owner, r2, y0, nrows, Wp);
            if (s + 3u * stride < n) rasterise_r2_padded(s_owner, r3, y0, nrows, Wp);
        }
        return;
    }
    for (uint32_t s = first; s < n; s += 4u * stride) {
        const uint2 r0 = st[s], r1 = st[min(s + stride, n - 1u)], r2 = st[min(s + 2u * stride, n - 1u)],
                    r3 = st[min(s + 3u * stride, n - 1u)];
        rasterise_one_padded(s_owner, r0, y0, nrows, Wp, radius, hw8, rowmask);
        if (s + stride < n) rasterise_one_padded(s_owner, r1, y0, nrows, Wp, radius, hw8, rowmask);
        if (s + 2u * stride < n) rasterise_one_padded(s_owner, r2, y0, nrows, Wp, radius, hw8, rowmask);
        if (s + 3u * stride < n) rasterise_one_padded(s_owner, r3, y0, nrows, Wp, radius, hw8, rowmask);
    }
}

__device__ __forceinline__ void rasterise_rest(uint32_t *s_owner, const uint2 *st, uint32_t first, uint32_t stride, uint32_t n,
                                               int y0, int nrows, int W, const Disc &disc)
{
    for (uint32_t s = first; s < n; s += 4u * stride) {
        const uint2 r0 = st[s], r1 = st[min(s + stride, n - 1u)], r2 = st[min(s + 2u * stride, n - 1u)],
                    r3 = st[min(s + 3u * stride, n - 1u)];
        rasterise_one(s_owner, r0, y0, nrows, W, disc);
        if (s + stride < n) rasterise_one(s_owner, r1, y0, nrows, W, disc);
        if (s + 2u * stride < n) rasterise_one(s_owner, r2, y0, nrows, W, disc);
        if (s + 3u * stride < n) rasterise_one(s_owner, r3, y0, nrows, W, disc);
    }
}

// per-byte (colour*a + source*(256-a) + 128) >> 8 on four packed bytes: even and odd bytes as two 16-bit lanes each
__device__ __forceinline__ uint32_t blend_bytes(uint32_t src, uint32_t col, uint32_t a)
{
    const uint32_t na = 256u - a;
    const uint32_t e = ((col & 0x00ff00ffu) * a + (src & 0x00ff00ffu) * na + 0x00800080u) >> 8;
    const uint32_t o = (((col >> 8) & 0x00ff00ffu) * a + ((src >> 8) & 0x00ff00ffu) * na + 0x00800080u) >> 8;
    return (e & 0x00ff00ffu) | ((o & 0x00ff00ffu) << 8);
}

template <bool ALPHA = false>
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
        c[k] = o[k] ? (((o[k] - 1u) & 1u) ? pal.c[1] : pal.c[0]) : 0u;     // select, not an indexed kernarg load
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
    if (!ALPHA) {                         // reference semantics: opaque overwrite (cv2.circle with a solid colour)
        d.x = (d.x & ~m0) | (v0 & m0);
        d.y = (d.y & ~m1) | (v1 & m1);
        d.z = (d.z & ~m2) | (v2 & m2);
        d.w = (d.w & ~m3) | (v3 & m3);
    } else {                              // extension: owned pixels = round(alpha*colour + (1-alpha)*source), once
        d.x = (d.x & ~m0) | (blend_bytes(d.x, v0, pal.alpha256) & m0);
        d.y = (d.y & ~m1) | (blend_bytes(d.y, v1, pal.alpha256) & m1);
        d.z = (d.z & ~m2) | (blend_bytes(d.z, v2, pal.alpha256) & m2);
        d.w = (d.w & ~m3) | (blend_bytes(d.w, v3, pal.alpha256) & m3);
    }
}

__device__ __forceinline__ u32x4 chunk_from_pixels(const uint32_t *c, uint32_t ph);

// per-byte (colour * a + source * (256 - a) + 128) >> 8 with a = coverage + (coverage >> 7), four bytes of a dword
__device__ __forceinline__ uint32_t blend_bytes_cov(uint32_t src, uint32_t col, uint32_t cov)
{
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t cv = (cov >> (8 * k)) & 255u, a = cv + (cv >> 7);
        const uint32_t s = (src >> (8 * k)) & 255u, c = (col >> (8 * k)) & 255u;
        out |= (((c * a + s * (256u - a) + 128u) >> 8) & 255u) << (8 * k);
    }
    return out;
}

// anti-aliased variant: owner cells are ((key + 1) << 8) | coverage; every owned pixel is blended once with its own coverage
__device__ __forceinline__ void patch_chunk_wu(u32x4 &d, const uint32_t *orow, uint32_t col, const Palette &pal)
{
    const uint32_t b0 = col * 16u, p0 = b0 / 3u, ph = b0 - p0 * 3u;
    uint32_t o[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = orow[p0 + k];
    if ((o[0] | o[1] | o[2] | o[3] | o[4] | o[5]) == 0u) return;
    uint32_t c[6], a[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        c[k] = o[k] ? ((((o[k] >> 8) - 1u) & 1u) ? pal.c[1] : pal.c[0]) : 0u;
        a[k] = (o[k] & 255u) * 0x010101u;                       // the pixel's coverage on each of its three bytes
    }
    const u32x4 v = chunk_from_pixels(c, ph), cv = chunk_from_pixels(a, ph);
    d.x = blend_bytes_cov(d.x, v.x, cv.x);
    d.y = blend_bytes_cov(d.y, v.y, cv.y);
    d.z = blend_bytes_cov(d.z, v.z, cv.z);
    d.w = blend_bytes_cov(d.w, v.w, cv.w);
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

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the
// band's source loads that are deliberately left in flight across the rasterisation.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ------------------------------------------------------------------------------------------
// Which band does workgroup L render?  (round 3)
// The bands of a launch are numbered in the order (frame, mosaic row of cameras, band, camera column, [column tile]):
// the `cols` cameras that share a mosaic row-band are neighbours, so consecutive items read consecutive source bytes and
// write R full mosaic rows.  Workgroups are dispatched round-robin over the 8 XCDs (observed: block L runs on XCD L % 8;
// /opt/skills/guides/MI355X_MICROARCH.md, "for speed only").  One function, three orders, chosen per launch by the host
// (cama_hip.hip: MapTuner / overlay_chunk_log2, with the measurements):
//   interleaved  (chunk_log2 = 0) workgroup L renders item L: all XCDs advance through ONE stream, each touching every page;
//   chunked      (1 .. 30) chunks of 2^k items dealt round-robin to the XCDs -- k = 5 is what small launches and the raw-frame
//                variants use: the same speed in every process, a little above the interleaved order;
//   contiguous   (31) XCD x renders items [x * ceil(T/8), (x+1) * ceil(T/8)): eight streams, each XCD touches an eighth of the
//                pages.  The fastest order for launches beyond ~2 GB in most processes (the headline's 40 x 1600x900 frames:
//                0.82 of 8 TB/s against 0.755 / 0.78 for the other two), the slowest in others (0.75): the process times it
//                against the chunked order on its own first launches.
// Either mapping is a bijection whatever the hardware's placement is: a different dispatch rule costs speed, never pixels.
// ------------------------------------------------------------------------------------------
// workgroup number L -> item
__device__ __forceinline__ bool xcd_item_of(const uint32_t L, const uint32_t T, const uint32_t chunk_log2, uint32_t &item)
{
    const uint32_t x = L & 7u, slot = L >> 3;
    if (chunk_log2 >= 31u) {                                 // one contiguous range per XCD; grid = 8 * ceil(T / 8)
        item = x * ((T + 7u) >> 3) + slot;
    } else {                                                 // chunks of K = 2^chunk_log2 items dealt round-robin to the XCDs;
        const uint32_t K = 1u << chunk_log2;                 // grid = 8 * K * ceil(T / (8 K))
        item = (((slot >> chunk_log2) << 3) + x) * K + (slot & (K - 1u));
    }
    return item < T;
}

// Diagnostic: which XCD does block L of a 1-D grid run on?  out[L] = HW_REG_XCC_ID (0..7).  The contiguous mapping above
// assumes L % 8 for speed; cama_probe_xcd_map() lets a caller (bench.py prints it) see what the box really does.
__global__ void k_probe_xcd(uint32_t *__restrict__ out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15u;
}

// n / d by multiply-high with magic = ceil(2^32 / d) (host): the estimate is the quotient or one more for every n < 2^32
// (excess n * (magic * d - 2^32) / (d * 2^32) < 1), hence one correction step; d = 1 has no 32-bit magic.
__device__ __forceinline__ uint32_t divmod_magic(const uint32_t n, const uint32_t d, const uint32_t magic, uint32_t &r)
{
    uint32_t q = d == 1u ? n : __umulhi(n, magic);
    r = n - q * d;
    if ((int32_t)r < 0) { q -= 1u; r += d; }
    return q;
}

struct BandId { uint32_t fl, c, b, tx; bool valid; };

// item -> (frame inside the launch, camera, band, column tile); TX = column tiles per band (1 for k_overlay)
__device__ __forceinline__ BandId decode_band(const OverlayArgs &a, const uint32_t T, const uint32_t TX, const uint32_t tx_magic,
                                              const uint32_t L = blockIdx.x)
{
    BandId id{0u, 0u, 0u, 0u, false};
    uint32_t item;
    if (!xcd_item_of(L, T, a.chunk_log2, item)) return id;
    const uint32_t cols = (uint32_t)a.cols, NB = (uint32_t)a.NB, C = (uint32_t)a.C, camrows = (C + cols - 1u) / cols;
    uint32_t cc, cr;
    const uint32_t q0 = TX == 1u ? item : divmod_magic(item, TX, tx_magic, id.tx);
    const uint32_t q1 = divmod_magic(q0, cols, a.cols_magic, cc);
    const uint32_t q2 = divmod_magic(q1, NB, a.nb_magic, id.b);
    id.fl = divmod_magic(q2, camrows, a.cr_magic, cr);
    id.c = cr * cols + cc;
    id.valid = id.c < C;                                  // ragged last camera row
    return id;
}

template <bool VEC, bool RESAMPLE, bool ALPHA, bool SEGS = false, bool WU = false>
__device__ __forceinline__ void overlay_band_at(const OverlayArgs &a, const uint32_t fl, const uint32_t f0, const uint32_t c,
                                                const uint32_t b, uint32_t *s_owner)
{
    const uint32_t NB = (uint32_t)a.NB, C = (uint32_t)a.C;
    const uint32_t fcl = fl * C + c;                     // (frame, camera) inside the scene's own frame tensor
    const uint32_t f = f0 + fl;
    const uint32_t fc = f * C + c;
    const uint32_t bin = fc * NB + b;
    const int y0 = (int)b * a.R;
    const int nrows = min(a.R, a.H - y0);
    const int W = a.W;
    const uint32_t n = a.counts[bin];

    // A stamped band fetches this thread's first stamp record and THEN issues its (first, normally only) batch of
    // 16-byte source loads, all before clearing / rasterising: the source chunks do not depend on the owner table, so
    // their HBM latency runs under the LDS work, and the stamp record -- issued first, VMEM returns in order -- can
    // be waited for without draining them.
    // (the record load is unconditional -- lanes without a stamp re-read the bin's last one, empty bins read
    // stamps[0] -- because a load inside a divergent branch is waited for right there)
    // (count, offsets: three independent scalar loads, one latency -- not count first and the offsets behind a branch)
    const uint32_t list0 = a.fc_base[fc] + a.bin_off[bin];
    const uint2 *st = a.stamps + (n ? (size_t)list0 : (size_t)0);
    const uint4 *st4 = reinterpret_cast<const uint4 *>(a.stamps) + (n ? (size_t)list0 : (size_t)0);   // (SEGS: 16-byte records)
    uint2 first = make_uint2(0u, 0u);
    uint4 first4 = make_uint4(0u, 0u, 0xffffffffu, 0u);
    if (SEGS) {
        first4 = st4[n ? min(threadIdx.x, n - 1u) : 0u];
        first = make_uint2(first4.x, first4.y);
    } else
        first = st[n ? min(threadIdx.x, n - 1u) : 0u];
    __builtin_amdgcn_sched_barrier(0);          // keep the record load ahead of the source loads (in-order vmcnt)

    const uint8_t *sband = a.src + ((size_t)fcl * a.H + y0) * (size_t)W * 3;     // (unused by RESAMPLE)
    constexpr int U = OVERLAY_UNROLL;
    const u32x4 *s16 = reinterpret_cast<const u32x4 *>(sband);
    const uint32_t nchunks = (uint32_t)nrows * a.cpr;
    u32x4 v[U];
    if (VEC && !RESAMPLE) {
        // unconditional (index clamped to the band's last chunk): loads under a divergent branch are waited for at
        // the end of the branch, which would serialise them
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = OVERLAY_LOAD(s16 + min(threadIdx.x + j * OVERLAY_BLOCK, nchunks - 1u));
    }
    // (issuing these loads BEFORE the count / offset / record chain was measured: 0.753 against 0.768 on the headline)

    // owner table: rows of Wp = W + 2 * radius cells, pixel x at cell x + radius (see rasterise_one_padded)
    const int rad = a.disc.radius, Wp = W + 2 * rad;
    if (n) {
        uint4 *o4 = reinterpret_cast<uint4 *>(s_owner);
        const int n4 = (nrows * Wp + 3) >> 2;
        for (int j = threadIdx.x; j < n4; j += OVERLAY_BLOCK) o4[j] = make_uint4(0, 0, 0, 0);
        lds_barrier();
        const uint32_t hw8 = (uint32_t)a.disc.hw4, rowmask = a.disc.rows;      // radius <= 7 (host-checked for this kernel)
        if (SEGS) {
            for (uint32_t sidx = threadIdx.x; sidx < n; sidx += OVERLAY_BLOCK) {
                const uint4 r = sidx == threadIdx.x ? first4 : st4[sidx];
                if (WU) rasterise_segment_wu_padded(s_owner, r, y0, nrows, Wp, rad);
                else rasterise_segment_padded(s_owner, r, y0, nrows, Wp, rad);
                // (the disc of a record that is here only because its segment crosses this band may lie outside it)
                const int v = (int)(r.x >> 16);
                if (v + rad >= y0 && v - rad < y0 + nrows) {
                    if (WU) rasterise_disc_wu_padded(s_owner, make_uint2(r.x, r.y), y0, nrows, Wp, rad, hw8, rowmask);
                    else if (disc_is_r2(rad, hw8, rowmask)) rasterise_r2_padded(s_owner, make_uint2(r.x, r.y), y0, nrows, Wp);
                    else rasterise_one_padded(s_owner, make_uint2(r.x, r.y), y0, nrows, Wp, rad, hw8, rowmask);
                }
            }
        } else {
            if (threadIdx.x < n) {
                if (disc_is_r2(rad, hw8, rowmask)) rasterise_r2_padded(s_owner, first, y0, nrows, Wp);
                else rasterise_one_padded(s_owner, first, y0, nrows, Wp, rad, hw8, rowmask);
            }
            rasterise_rest_padded(s_owner, st, threadIdx.x + OVERLAY_BLOCK, OVERLAY_BLOCK, n, y0, nrows, Wp, rad, hw8, rowmask);
        }
        lds_barrier();
    }

    uint8_t *dcell = a.mosaic + (size_t)fl * a.mosaic_frame_bytes +
                     ((size_t)(c / (uint32_t)a.cols) * a.H + y0) * a.mosaic_row_bytes +
                     (size_t)(c % (uint32_t)a.cols) * W * 3;

    if (RESAMPLE) {
        // The source is the raw sensor frame of (f, c): every destination pixel is the fixed-point bilinear blend of
        // its 2x2 source taps (cv2.remap semantics), computed here instead of being read from a pre-resized frame.
        // A 16-byte chunk overlaps 6 destination pixels; adjacent lanes own adjacent chunks, so their taps are
        // adjacent in the raw frame.  Raw bytes are read once from HBM (re-reads hit L1/L2), resized frames never
        // exist in memory.
        const size_t raw_frame = (size_t)a.H0 * a.W0 * 3;
        const uint8_t *raw = a.src + (size_t)fcl * raw_frame;
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
            if (n) patch_chunk(v, s_owner + row * Wp + rad, col, a.pal);
            u32x4 *drow = reinterpret_cast<u32x4 *>(dcell + (size_t)row * a.mosaic_row_bytes);
            OVERLAY_STORE(v, drow + col);
        }
        return;
    }

    if (VEC) {
        // the band is one contiguous byte range in src: chunk j of the band is src16[j]
        for (uint32_t base = threadIdx.x; base < nchunks; base += OVERLAY_BLOCK * U) {
            if (base != threadIdx.x) {              // later batches (bands wider than OVERLAY_BLOCK * U chunks)
#pragma unroll
                for (int j = 0; j < U; ++j) v[j] = OVERLAY_LOAD(s16 + min(base + j * OVERLAY_BLOCK, nchunks - 1u));
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const uint32_t idx = base + j * OVERLAY_BLOCK;
                if (idx < nchunks) {
                    const uint32_t row = __umulhi(idx, a.cpr_magic);
                    const uint32_t col = idx - row * a.cpr;
                    if (n) {
                        if (WU) patch_chunk_wu(v[j], s_owner + row * Wp + rad, col, a.pal);
                        else patch_chunk<ALPHA>(v[j], s_owner + row * Wp + rad, col, a.pal);
                    }
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
                const uint32_t o = s_owner[row * Wp + rad + x];
                if (o) {
                    const uint32_t col = ((o - 1u) & 1u) ? a.pal.c[1] : a.pal.c[0];
                    const uint32_t srcw = (uint32_t)b0 | ((uint32_t)b1 << 8) | ((uint32_t)b2 << 16);
                    const uint32_t w = ALPHA ? blend_bytes(srcw, col, a.pal.alpha256) : col;
                    b0 = (uint8_t)w; b1 = (uint8_t)(w >> 8); b2 = (uint8_t)(w >> 16);
                }
            }
            uint8_t *d = dcell + (size_t)row * a.mosaic_row_bytes + (size_t)x * 3;
            d[0] = b0; d[1] = b1; d[2] = b2;
        }
    }
}

template <bool VEC, bool RESAMPLE, bool ALPHA = false, bool SEGS = false, bool WU = false>
__global__ __launch_bounds__(OVERLAY_BLOCK) void k_overlay(OverlayArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_owner[];  // R x (W + 2 radius), used only by stamped bands
    const BandId id = decode_band(a, a.items, 1u, 0u);
    if (id.valid) overlay_band_at<VEC, RESAMPLE, ALPHA, SEGS, WU>(a, id.fl, (uint32_t)a.f0, id.c, id.b, s_owner);
}


// The plain overlay under its own name, for cama_overlay_probe (timing candidate allocations of long-lived buffers): the same
// body, but a workload's rocprofv3 --stats keeps the probe launches apart from k_overlay's.
__global__ __launch_bounds__(OVERLAY_BLOCK) void k_overlay_probe(OverlayArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_owner[];
    const BandId id = decode_band(a, a.items, 1u, 0u);
    if (id.valid) overlay_band_at<true, false, false, false>(a, id.fl, (uint32_t)a.f0, id.c, id.b, s_owner);
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
                                                                 const int2 *__restrict__ tile_bytes, int TX, int Wt,
                                                                 uint32_t tx_magic)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const uint32_t cols = (uint32_t)a.cols, NB = (uint32_t)a.NB, C = (uint32_t)a.C;
    const BandId id = decode_band(a, a.items, (uint32_t)TX, tx_magic);
    if (!id.valid) return;
    const uint32_t f = id.fl, c = id.c, b = id.b, tx = id.tx;
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
                const int hw = disc_halfwidth(a.disc, abs(y - v));
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
        const int hw = disc_halfwidth(disc, abs(dy));
        if (hw < 0) continue;
        const int xlo = max(u - hw, 0), xhi = min(u + hw, W - 1);
        for (int x = xlo; x <= xhi; ++x) atomicMax(&owner[(size_t)y * W + x], val);
    }
}

// EXTENSION (no reference semantics: the reference draws one disc per point and nothing between them):
// point k with link[k] != 0 is also joined to point k - 1 by a one-pixel-wide 8-connected Bresenham segment between the two
// truncated pixel positions, drawn under point k's key (so it belongs to point k's instance and is covered by whatever a
// later point draws).  One thread per segment; the integer error recurrence below is the definition (oracle:
// oracle_line_bresenham, the same statements), every octant and direction, both end pixels included.
__global__ __launch_bounds__(BLOCK) void k_segments_global(const double *__restrict__ vu, const uint8_t *__restrict__ colour,
                                                           const uint8_t *__restrict__ link, int64_t n,
                                                           uint32_t *__restrict__ owner, int H, int W)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n || i == 0 || !link[i]) return;
    int y0 = (int)vu[2 * (i - 1)], x0 = (int)vu[2 * (i - 1) + 1];
    const int y1 = (int)vu[2 * i], x1 = (int)vu[2 * i + 1];
    const uint32_t val = ((((uint32_t)i) << 1) | (uint32_t)(colour[i] & 1)) + 1u;
    const int dx = abs(x1 - x0), sx = x0 < x1 ? 1 : -1;
    const int dy = -abs(y1 - y0), sy = y0 < y1 ? 1 : -1;
    int err = dx + dy;
    for (int guard = 0; guard < 4 * 65536; ++guard) {                  // (a segment has at most W + H pixels)
        if ((unsigned)x0 < (unsigned)W && (unsigned)y0 < (unsigned)H) atomicMax(&owner[(size_t)y0 * W + x0], val);
        if (x0 == x1 && y0 == y1) break;
        const int e2 = 2 * err;
        if (e2 >= dy) { err += dy; x0 += sx; }
        if (e2 <= dx) { err += dx; y0 += sy; }
    }
}

// ... the anti-aliased variant of the two kernels above for the one-image path (cama_stamp_polylines_wu): owner cells are
// ((key + 1) << 8) | coverage, discs claim 255, segments the Wu coverages of wu_line_claims (oracle/cama_oracle.c).
__global__ __launch_bounds__(BLOCK) void k_stamp_global_wu(const double *__restrict__ vu, const uint8_t *__restrict__ colour,
                                                           const uint8_t *__restrict__ link, int64_t n,
                                                           uint32_t *__restrict__ owner, int H, int W, Disc disc)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int v = (int)vu[2 * i], u = (int)vu[2 * i + 1];
    const uint32_t key1 = (((((uint32_t)i) << 1) | (uint32_t)(colour[i] & 1)) + 1u) << 8;
    const auto claim = [&](int x, int y, uint32_t val) {
        if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H && (val & 255u)) atomicMax(&owner[(size_t)y * W + x], val);
    };
    for (int dy = -disc.radius; dy <= disc.radius; ++dy) {
        const int hw = disc_halfwidth(disc, abs(dy));
        for (int dx = -hw; dx <= hw; ++dx) claim(u + dx, v + dy, key1 | 255u);
    }
    if (i == 0 || !link || !link[i]) return;
    const int yy0 = (int)vu[2 * (i - 1)], x0 = (int)vu[2 * (i - 1) + 1];
    if (yy0 == v && x0 == u) return;
    const bool steep = abs(v - yy0) > abs(u - x0);
    int a0 = steep ? yy0 : x0, b0 = steep ? x0 : yy0, a1 = steep ? v : u, b1 = steep ? u : v;
    if (a0 > a1) { int t = a0; a0 = a1; a1 = t; t = b0; b0 = b1; b1 = t; }
    const int da = a1 - a0, num = (b1 - b0) * 65536;
    int grad = num / da;
    if (num % da != 0 && num < 0) grad -= 1;
    for (int j = 0; j <= da; ++j) {
        const int y = b0 * 65536 + grad * j, row = y >> 16, a = a0 + j;
        const uint32_t f = ((uint32_t)y & 0xffffu) >> 8;
        if (steep) { claim(row, a, key1 | (255u - f)); claim(row + 1, a, key1 | f); }
        else { claim(a, row, key1 | (255u - f)); claim(a, row + 1, key1 | f); }
    }
}

__global__ __launch_bounds__(BLOCK) void k_apply_owner_wu(const uint32_t *__restrict__ owner, uint8_t *__restrict__ image,
                                                          int64_t npix, Palette pal)
{
    const int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p >= npix) return;
    const uint32_t o = owner[p];
    if (!o) return;
    const uint32_t col = (((o >> 8) - 1u) & 1u) ? pal.c[1] : pal.c[0], cov = o & 255u, a = cov + (cov >> 7);
#pragma unroll
    for (int k = 0; k < 3; ++k)
        image[3 * p + k] = (uint8_t)((((col >> (8 * k)) & 255u) * a + (uint32_t)image[3 * p + k] * (256u - a) + 128u) >> 8);
}

__global__ __launch_bounds__(BLOCK) void k_apply_owner(const uint32_t *__restrict__ owner,
                                                       uint8_t *__restrict__ image, int64_t npix, Palette pal)
{
    const int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (p >= npix) return;
    const uint32_t o = owner[p];
    if (!o) return;
    const uint32_t col = ((o - 1u) & 1u) ? pal.c[1] : pal.c[0];
    image[3 * p] = (uint8_t)col;
    image[3 * p + 1] = (uint8_t)(col >> 8);
    image[3 * p + 2] = (uint8_t)(col >> 16);
}
