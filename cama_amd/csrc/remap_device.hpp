// remap_device.hpp -- part of libcama_hip.so (included by cama_hip.hip inside its anonymous namespace).
// cv2.remap building blocks shared by the resample kernel and the raw-frame overlays.
#pragma once

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
