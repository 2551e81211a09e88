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

// four packed pixels (b | g<<8 | r<<16) -> the 12 bytes they occupy, as three dwords
__device__ __forceinline__ void raw35_pack4(const uint32_t *c, uint32_t *o)
{
    o[0] = c[0] | (c[1] << 24);
    o[1] = (c[1] >> 8) | (c[2] << 16);
    o[2] = (c[2] >> 16) | (c[3] << 8);
}

// ------------------------------------------------------------------------------------------
// Packed arithmetic (round 3).  PMC showed the first version VALU-bound (76 % of SIMD issue cycles at 5 waves per SIMD:
// ~8.5 VALU instructions per output byte value -- three v_dot4 for the horizontal taps of both rows, two 24-bit multiplies,
// the rounding add, the shift, the byte packing).  The blend is exact integer arithmetic without intermediate rounding,
// v = (sum_ij wv_i wh_j p_ij + 512) >> 10, so the order is free: VERTICAL FIRST on two taps at a time in packed 16-bit
// lanes (q = wt*a + wb*b <= 32*255 fits 16 bits), then the horizontal pair with one v_dot2_u32_u16:
//     A = v_perm(row0 dwords) = [a_e, 0, a_e+3, 0]        B = the same bytes of row 1
//     Q = v_pk_mad_u16(A, wt, v_pk_mul_lo_u16(B, wb))      = [q_e, q_e+3]
//     r = v_dot2_u32_u16(Q, [64 wl, 64 wr], 32768)         = 64 (S + 512): the result (S + 512) >> 10 is BYTE 2 of r
// -- 5 instructions per two-tap value; the one-tap values (phase 0 columns, weights 32 | 0) go two at a time through
//     Q' = v_pk_mad_u16(A, 8 wt, v_pk_mad_u16(B, 8 wb, 128)) = 8 q + 128: (q + 16) >> 5 is the HIGH byte of each lane
// -- 2 instructions per value; and because every result sits on a byte boundary, the 36 output bytes are gathered with
// v_perm_b32 (3 per output dword) instead of shift / or chains.  ~170 instead of ~350 VALU instructions per 12-pixel unit.
// ------------------------------------------------------------------------------------------
typedef unsigned short raw35_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t raw35_pk_mul(uint32_t a, uint32_t w)
{
    return __builtin_bit_cast(uint32_t, (raw35_u16x2)(__builtin_bit_cast(raw35_u16x2, a) * __builtin_bit_cast(raw35_u16x2, w)));
}
__device__ __forceinline__ uint32_t raw35_pk_mad(uint32_t a, uint32_t w, uint32_t c)
{
    return __builtin_bit_cast(uint32_t, (raw35_u16x2)(__builtin_bit_cast(raw35_u16x2, a) * __builtin_bit_cast(raw35_u16x2, w) +
                                                      __builtin_bit_cast(raw35_u16x2, c)));
}
// v_perm_b32: result byte i = byte sel_i of the 8 bytes {hi, lo} (0..3 = lo, 4..7 = hi), 0x0c = 0x00
__device__ __forceinline__ uint32_t raw35_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// source byte offset (inside the unit's 60 bytes) of the LEFT tap of destination pixel K, channel CH; the right tap is 3 on
constexpr int raw35_e(int K, int CH) { return 3 * (5 * (K / 3) + (K % 3 == 0 ? 0 : K % 3 == 1 ? 1 : 3)) + CH; }

// [d_E1, 0, d_E2, 0] from the unit's dwords
template <int E1, int E2>
__device__ __forceinline__ uint32_t raw35_gather2(const uint32_t (&d)[15])
{
    constexpr int w1 = E1 >> 2, w2 = E2 >> 2;
    static_assert(w1 < 15 && w2 < 15, "tap outside the unit");
    constexpr uint32_t sel = (uint32_t)(E1 & 3) | (0x0cu << 8) | ((uint32_t)((w1 == w2 ? 0 : 4) + (E2 & 3)) << 16) | (0x0cu << 24);
    return raw35_perm(d[w2], d[w1], sel);
}

// Values of a unit, as registers whose bytes hold results: reg index + byte position of output byte m = 3 K + CH.
//   two-tap pixels (K % 3 != 0): one register per (K, CH), value in byte 2           -> regs 0 .. 23
//   one-tap pixels (K % 3 == 0: K = 0, 3, 6, 9), 12 values, two per register (bytes 1 and 3) -> regs 24 .. 29,
//   paired in output order (m = 0,1 | 2,9 | 10,11 | 18,19 | 20,27 | 28,29) so that neighbours in a dword share a register
constexpr int raw35_two_index(int K, int CH) { return ((K / 3) * 2 + (K % 3 - 1)) * 3 + CH; }
constexpr int raw35_one_order(int K, int CH) { return (K / 3) * 3 + CH; }          // 0 .. 11 in output order
constexpr int raw35_reg_of(int m)
{
    const int K = m / 3, CH = m % 3;
    return K % 3 ? raw35_two_index(K, CH) : 24 + raw35_one_order(K, CH) / 2;
}
constexpr int raw35_byte_of(int m)
{
    const int K = m / 3, CH = m % 3;
    return K % 3 ? 2 : 1 + 2 * (raw35_one_order(K, CH) % 2);
}

template <int K, int CH>
__device__ __forceinline__ uint32_t raw35_two_tap(const uint32_t (&d0)[15], const uint32_t (&d1)[15], uint32_t wt2, uint32_t wb2)
{
    constexpr int E = raw35_e(K, CH);
    constexpr uint32_t wl = K % 3 == 1 ? 11u : 21u, wr = 32u - wl;
    const uint32_t A = raw35_gather2<E, E + 3>(d0), B = raw35_gather2<E, E + 3>(d1);
    const uint32_t Q = raw35_pk_mad(A, wt2, raw35_pk_mul(B, wb2));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(raw35_u16x2, Q), __builtin_bit_cast(raw35_u16x2, (64u * wl) | ((64u * wr) << 16)),
                                  32768u, false);
}

template <int J>        // one-tap pair J (output order): values j = 2 J and 2 J + 1 of the 12 one-tap values
__device__ __forceinline__ uint32_t raw35_one_tap_pair(const uint32_t (&d0)[15], const uint32_t (&d1)[15], uint32_t wt8, uint32_t wb8)
{
    constexpr int j1 = 2 * J, j2 = 2 * J + 1;
    constexpr int E1 = raw35_e((j1 / 3) * 3, j1 % 3), E2 = raw35_e((j2 / 3) * 3, j2 % 3);
    const uint32_t A = raw35_gather2<E1, E2>(d0), B = raw35_gather2<E1, E2>(d1);
    return raw35_pk_mad(A, wt8, raw35_pk_mad(B, wb8, 0x00800080u));
}

template <int I>
__device__ __forceinline__ uint32_t raw35_value(const uint32_t (&d0)[15], const uint32_t (&d1)[15], uint32_t wt2, uint32_t wb2,
                                            uint32_t wt8, uint32_t wb8)
{
    if constexpr (I < 24)
        return raw35_two_tap<(I / 3 / 2) * 3 + (I / 3) % 2 + 1, I % 3>(d0, d1, wt2, wb2);
    else
        return raw35_one_tap_pair<I - 24>(d0, d1, wt8, wb8);
}

template <int... I>
__device__ __forceinline__ void raw35_values(const uint32_t (&d0)[15], const uint32_t (&d1)[15], uint32_t wt, uint32_t wb,
                                             uint32_t (&r)[30], std::integer_sequence<int, I...>)
{
    const uint32_t wt2 = wt | (wt << 16), wb2 = wb | (wb << 16), wt8 = wt2 << 3, wb8 = wb2 << 3;
    // I = 0 .. 23: two-tap register I <-> (K, CH) with raw35_two_index(K, CH) == I;  I = 24 .. 29: one-tap pairs
    ((r[I] = raw35_value<I>(d0, d1, wt2, wb2, wt8, wb8)), ...);
}

// output dword J of the unit (bytes 4 J .. 4 J + 3) gathered from the value registers
template <int J>
__device__ __forceinline__ uint32_t raw35_out_dword(const uint32_t (&r)[30])
{
    constexpr int r0 = raw35_reg_of(4 * J), r1 = raw35_reg_of(4 * J + 1), r2 = raw35_reg_of(4 * J + 2), r3 = raw35_reg_of(4 * J + 3);
    constexpr uint32_t b0 = raw35_byte_of(4 * J), b1 = raw35_byte_of(4 * J + 1), b2 = raw35_byte_of(4 * J + 2),
                       b3 = raw35_byte_of(4 * J + 3);
    uint32_t t = raw35_perm(r[r1], r[r0], b0 | ((4u + b1) << 8) | (0x0cu << 16) | (0x0cu << 24));
    if constexpr (r2 == r3) {
        return raw35_perm(r[r2], t, 0u | (1u << 8) | ((4u + b2) << 16) | ((4u + b3) << 24));
    } else {
        t = raw35_perm(r[r2], t, 0u | (1u << 8) | ((4u + b2) << 16) | (0x0cu << 24));
        return raw35_perm(r[r3], t, 0u | (1u << 8) | (2u << 16) | ((4u + b3) << 24));
    }
}

template <int... J>
__device__ __forceinline__ void raw35_gather_out(const uint32_t (&r)[30], uint32_t (&o)[9], std::integer_sequence<int, J...>)
{
    ((o[J] = raw35_out_dword<J>(r)), ...);
}

// A stamped band's owner table lives INSIDE the staging area, right behind the band's output rows (owner_off dwords; the
// staging area is always big enough at a 3:5 scale: rows * 1.25 W >= R * 1.75 W dwords, host-checked): it is built after
// every unit has read its taps, so a stamped band pays two more barriers and its rasterisation no longer overlaps the
// source loads, but every workgroup needs 38 KB of LDS instead of 54 KB -- 4 instead of 3 per CU: 140.0 -> 145.5 k
// frames/s at N = 10^4, 117.7 -> 123.5 k at N = 10^5 (same box).
// Column tiles (round 3): a band is cut into TX tiles of upr / TX units, one workgroup each.  The ablations behind it
// (tools/ab, 960x540 from 1600x900, one box): stage -> store without any arithmetic 223 us, + tap reads and the transpose
// 227 us, + the arithmetic 249 us -- so the kernel is neither at the memory system's limit for this traffic shape nor short
// of VALU throughput (VALU is ~1/3 busy): with 38 KB of LDS per band a CU holds 4 workgroups = 5 waves per SIMD, too few to
// hide a workgroup's compute phase behind the others' loads.  Half a band needs half the staging: 8 workgroups per CU.
__device__ __forceinline__ void raw35_rasterise_tile(uint32_t *s_owner, const uint2 r, int y0, int nrows, int x_first, int Wt,
                                                     const Disc &disc)
{
    const int u = (int)(r.x & 0xffffu), v = (int)(r.x >> 16);
    const uint32_t val = r.y + 1u;
    const int ylo = max(v - disc.radius, y0), yhi = min(v + disc.radius, y0 + nrows - 1);
    for (int y = ylo; y <= yhi; ++y) {
        const int hw = disc_halfwidth(disc, abs(y - v));
        if (hw < 0) continue;
        const int xlo = max(u - hw, x_first), xhi = min(u + hw, x_first + Wt - 1);
        uint32_t *row = s_owner + (y - y0) * Wt - x_first;
        for (int x = xlo; x <= xhi; ++x) atomicMax(&row[x], val);
    }
}

// One band's share of the work, in three phases, so that a workgroup can have the NEXT band's source loads in flight while
// it blends the current one (k_overlay_raw35 with bands_per_wg = 2).
struct Raw35Band {
    uint32_t b, fc, n;
    int y0, nrows;
    const uint2 *st;
    uint2 first;
    int2 br;                    // {first source row, number of source rows}
    const u32x4 *g;
    uint32_t nsrc;
};
struct Raw35Tile {
    int upr, Wt, x_first, W0, owner_off;
    uint32_t row_dwords, src_row_dwords, src_pitch16, cpt, cpt_magic, tx;
};

// phase 1: scalar loads (count, list offset, source rows), the band's first stamp record, then the source loads
template <int U, bool LOAD = true>
__device__ __forceinline__ void raw35_issue(const OverlayArgs &a, const int2 *__restrict__ band_rows, const Raw35Tile &t,
                                            const uint32_t f, const uint32_t c, const uint32_t b, Raw35Band &k, u32x4 (&v)[U])
{
    const uint32_t NB = (uint32_t)a.NB, C = (uint32_t)a.C;
    k.b = b;
    k.fc = f * C + c;
    const uint32_t bin = k.fc * NB + b;
    k.y0 = (int)b * a.R;
    k.nrows = min(a.R, a.H - k.y0);
    k.n = a.counts[bin];
    const uint32_t list0 = a.fc_base[k.fc] + a.bin_off[bin];          // (unconditional: three parallel scalar loads)
    k.st = a.stamps + (k.n ? (size_t)list0 : (size_t)0);
    // the band's first stamp record before the source loads (VMEM returns in order; see k_overlay)
    k.first = k.st[k.n ? min(threadIdx.x, k.n - 1u) : 0u];
    __builtin_amdgcn_sched_barrier(0);
    k.br = band_rows[c * NB + b];
    k.g = reinterpret_cast<const u32x4 *>(a.src + ((size_t)k.fc * a.H0 + k.br.x) * (size_t)a.W0 * 3) + t.tx * t.cpt;
    k.nsrc = (uint32_t)k.br.y * t.cpt;
    // chunk idx of the tile = (source row idx / cpt, chunk idx % cpt) (cpt_magic = ceil(2^32 / cpt) from the host: exact for
    // idx < 2^16); one tile per band: pitch == cpt, one contiguous range
    if constexpr (LOAD) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint32_t idx = min(threadIdx.x + j * blockDim.x, k.nsrc - 1u);
            const uint32_t r = __umulhi(idx, t.cpt_magic);
            v[j] = OVERLAY_LOAD(k.g + (size_t)r * t.src_pitch16 + (idx - r * t.cpt));
        }
    }
}

// phase 2: the staged rows into LDS (same layout as in memory)
template <int U>
__device__ __forceinline__ void raw35_stage(const Raw35Tile &t, const Raw35Band &k, const u32x4 (&v)[U], uint32_t *s_stage)
{
    u32x4 *s16 = reinterpret_cast<u32x4 *>(s_stage);
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const uint32_t idx = threadIdx.x + j * blockDim.x;
        if (idx < k.nsrc) s16[idx] = v[j];
    }
    for (uint32_t idx = threadIdx.x + U * blockDim.x; idx < k.nsrc; idx += blockDim.x) {
        const uint32_t r = __umulhi(idx, t.cpt_magic);
        s16[idx] = OVERLAY_LOAD(k.g + (size_t)r * t.src_pitch16 + (idx - r * t.cpt));
    }
    lds_barrier();
}

// phase 3: taps -> blend -> (stamps) -> transpose through the dead staging area -> 16-byte stores.  Only LDS barriers: the
// next band's source loads may be in flight.
// nthreads = the threads that take part (the whole block, or -- k_overlay_raw35_ws -- its blend waves)
__device__ __forceinline__ void raw35_finish(const OverlayArgs &a, const uint2 *__restrict__ vrows, const Raw35Tile &t,
                                             const uint32_t f, const uint32_t c, const Raw35Band &k, uint32_t *s_stage,
                                             const bool more_follows, const uint32_t nthreads)
{
    const uint32_t cols = (uint32_t)a.cols;
    uint32_t *s_owner = s_stage + t.owner_off;                         // [R * Wt], stamped bands only
    const int y0 = k.y0, nrows = k.nrows, W = a.W, Wt = t.Wt, x_first = t.x_first;
    const uint32_t n = k.n;
    // one 12-pixel unit per thread (the block covers the band: items <= blockDim, host-checked)
    const uint32_t items = (uint32_t)nrows * (uint32_t)t.upr;
    const bool active = threadIdx.x < items;
    const uint32_t item = active ? threadIdx.x : items - 1u;
    const uint32_t row = item / (uint32_t)t.upr, u = item - row * (uint32_t)t.upr;
    const uint2 vr = vrows[(size_t)c * a.H + y0 + (int)row];           // {r0 | r1 << 16, wt | wb << 8}
    const uint32_t wt = vr.y & 0xffu, wb = (vr.y >> 8) & 0xffu;
    // lane stride 15 dwords: odd, so the 32 banks are hit once each per half wave
    const uint32_t *p0 = s_stage + ((vr.x & 0xffffu) - (uint32_t)k.br.x) * t.src_row_dwords + u * 15u;
    const uint32_t *p1 = s_stage + ((vr.x >> 16) - (uint32_t)k.br.x) * t.src_row_dwords + u * 15u;
    uint32_t d0[15], d1[15];
#pragma unroll
    for (int j = 0; j < 15; ++j) d0[j] = p0[j];
#pragma unroll
    for (int j = 0; j < 15; ++j) d1[j] = p1[j];
    if (n) {                                                           // (workgroup-uniform)
        lds_barrier();                                                 // every unit holds its taps: staging is dead
        uint4 *o4 = reinterpret_cast<uint4 *>(s_owner);
        const int n4 = (nrows * Wt + 3) >> 2;
        for (int j = threadIdx.x; j < n4; j += (int)nthreads) o4[j] = make_uint4(0, 0, 0, 0);
        lds_barrier();
        if (threadIdx.x < n) raw35_rasterise_tile(s_owner, k.first, y0, nrows, x_first, Wt, a.disc);
        for (uint32_t sidx = threadIdx.x + nthreads; sidx < n; sidx += nthreads)
            raw35_rasterise_tile(s_owner, k.st[sidx], y0, nrows, x_first, Wt, a.disc);
        lds_barrier();
    }
    uint32_t o[9];
    {
        uint32_t r[30];
        raw35_values(d0, d1, wt, wb, r, std::make_integer_sequence<int, 30>{});
        raw35_gather_out(r, o, std::make_integer_sequence<int, 9>{});
    }
    if (n) {
        // stamped band: owned pixels take the palette colour -- colour and mask streams of the unit's 12 pixels, packed
        // like the output, then one select per dword
        const uint4 *orow = reinterpret_cast<const uint4 *>(s_owner + row * (uint32_t)Wt + u * 12u);
        uint32_t cv[12], cm[12];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint4 ow = orow[j];
            const uint32_t w4[4] = {ow.x, ow.y, ow.z, ow.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cm[4 * j + q] = w4[q] ? 0x00ffffffu : 0u;
                cv[4 * j + q] = w4[q] ? (((w4[q] - 1u) & 1u) ? a.pal.c[1] : a.pal.c[0]) : 0u;
            }
        }
        uint32_t V[9], M[9];
        raw35_pack4(cv, V);     raw35_pack4(cv + 4, V + 3); raw35_pack4(cv + 8, V + 6);
        raw35_pack4(cm, M);     raw35_pack4(cm + 4, M + 3); raw35_pack4(cm + 8, M + 6);
#pragma unroll
        for (int j = 0; j < 9; ++j) o[j] = (o[j] & ~M[j]) | V[j];
    }
    lds_barrier();                                                     // every unit has read its taps: staging is dead
    if (active) {
        uint32_t *dst = s_stage + row * t.row_dwords + u * 9u;         // R * row_dwords <= staging size (host-checked)
#pragma unroll
        for (int j = 0; j < 9; ++j) dst[j] = o[j];
    }
    lds_barrier();
    uint8_t *dcell = a.mosaic + (size_t)f * a.mosaic_frame_bytes +
                     ((size_t)(c / cols) * a.H + y0) * a.mosaic_row_bytes + (size_t)(c % cols) * W * 3 + (size_t)x_first * 3;
    const uint32_t cpr = t.row_dwords >> 2;                             // 16-byte chunks per destination tile row
    const uint32_t nchunks = (uint32_t)nrows * cpr;
    uint32_t r = threadIdx.x / cpr, col = threadIdx.x - r * cpr;        // one division, then (row, chunk) advance by blockDim
    const uint32_t dr = nthreads / cpr, dc = nthreads - dr * cpr;
    for (uint32_t idx = threadIdx.x; idx < nchunks; idx += nthreads) {
        const u32x4 w = reinterpret_cast<const u32x4 *>(s_stage + r * t.row_dwords)[col];
        OVERLAY_STORE(w, reinterpret_cast<u32x4 *>(dcell + (size_t)r * a.mosaic_row_bytes) + col);
        r += dr;
        col += dc;
        if (col >= cpr) { col -= cpr; ++r; }
    }
    if (more_follows) lds_barrier();                                   // the next band overwrites the staging area
}

// bands_per_wg = 1: one band (x one column tile) per workgroup, items = F * camera rows * cols * NB * TX.
// bands_per_wg = 2 (round 3): a workgroup renders bands 2p and 2p + 1 of one camera and issues the second band's source loads
// right after the first band is in LDS, so they fly during the first band's blend, transpose and stores:
// items = F * camera rows * cols * ceil(NB / 2) * TX, nbx_magic = ceil(2^32 / ceil(NB / 2)).
template <int bands_per_wg>
__global__ __launch_bounds__(RAW35_MAX_BLOCK) void k_overlay_raw35(OverlayArgs a, const uint2 *__restrict__ vrows,
                                                                   const int2 *__restrict__ band_rows, int upr,
                                                                   int max_src_rows, int owner_off, int TX,
                                                                   uint32_t tx_magic, uint32_t cpt_magic, uint32_t nbx_magic)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    // same item order as k_overlay -- (frame, mosaic row of cameras, band [pair], camera column), column tile innermost -- and
    // the same workgroup -> item mapping (overlay_kernels.hpp: xcd_item_of)
    const uint32_t cols = (uint32_t)a.cols, NB = (uint32_t)a.NB, C = (uint32_t)a.C, camrows = (C + cols - 1u) / cols;
    const uint32_t NBx = bands_per_wg == 2 ? (NB + 1u) >> 1 : NB;
    uint32_t item;
    if (!xcd_item_of(blockIdx.x, a.items, a.chunk_log2, item)) return;
    uint32_t tx = 0, cc, bx, cr;
    const uint32_t q0 = TX == 1 ? item : divmod_magic(item, (uint32_t)TX, tx_magic, tx);
    const uint32_t q1 = divmod_magic(q0, cols, a.cols_magic, cc);
    const uint32_t q2 = divmod_magic(q1, NBx, nbx_magic, bx);
    const uint32_t f = divmod_magic(q2, camrows, a.cr_magic, cr);
    const uint32_t c = cr * cols + cc;
    if (c >= C) return;
    (void)max_src_rows;
    Raw35Tile t;
    t.upr = upr; t.Wt = upr * 12; t.x_first = (int)tx * t.Wt; t.W0 = a.W0; t.owner_off = owner_off;
    t.row_dwords = (uint32_t)upr * 9u;                                  // destination dwords per tile row (multiple of 4)
    t.src_row_dwords = (uint32_t)upr * 15u;                             // source dwords per tile row (multiple of 4)
    t.src_pitch16 = (uint32_t)a.W0 * 3u / 16u;                          // 16-byte chunks per raw row (W0*3 % 16 == 0)
    t.cpt = t.src_row_dwords >> 2; t.cpt_magic = cpt_magic; t.tx = tx;
    uint32_t *s_stage = s_dyn;                                          // [max_src_rows * src_row_dwords], later the output
    constexpr int U = RAW35_STAGE_UNROLL;
    if constexpr (bands_per_wg != 2) {
        Raw35Band k;
        constexpr bool lds_direct = true;
        // (only when the staged rows are contiguous in memory, i.e. the raw row pitch equals the staged row: W0 * 3 == 5 * W
        // bytes; a wider sensor row -- 5 * W / 3 < W0, which the plan accepts -- takes the register-staged path with its
        // src_pitch16 stride)
        if (lds_direct && TX == 1 && t.src_pitch16 == t.cpt) {
            // source rows straight into LDS (gfx950's global_load_lds_dwordx4: a wave's 64 x 16 bytes land at M0 + 16 lane):
            // no VGPR staging, no ds_write -- with one tile per band the staged rows are one contiguous range of memory and
            // keep their layout.  Measured at 960x540 against the register-staged version (-DRAW35_NO_LDS_DIRECT),
            // alternating, three each on one box: 141.5 -> 143.7 k frames/s with the non-temporal hint, 140.5 k without.
            u32x4 none[1];
            raw35_issue<1, false>(a, band_rows, t, f, c, bx, k, none);
            for (uint32_t base = threadIdx.x & ~63u; base < k.nsrc; base += blockDim.x) {
                const uint32_t idx = base + (threadIdx.x & 63u);
                if (idx < k.nsrc)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(k.g + idx),
                                                     (__attribute__((address_space(3))) void *)(s_stage + base * 4u), 16, 0,
                                                     2 /* nt */);
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);                         // vmcnt(0)
            lds_barrier();
        } else {
            u32x4 v[U];
            raw35_issue<U>(a, band_rows, t, f, c, bx, k, v);
            if (!k.nrows) return;
            raw35_stage<U>(t, k, v, s_stage);
        }
        raw35_finish(a, vrows, t, f, c, k, s_stage, false, blockDim.x);
    } else {
    const uint32_t b0 = 2u * bx, b1 = b0 + 1u;
    const bool two = b1 < NB;
    Raw35Band kA, kB;
    u32x4 vA[U], vB[U];
    raw35_issue<U>(a, band_rows, t, f, c, b0, kA, vA);
    raw35_stage<U>(t, kA, vA, s_stage);
    if (two) raw35_issue<U>(a, band_rows, t, f, c, b1, kB, vB);       // in flight during the first band's finish
    raw35_finish(a, vrows, t, f, c, kA, s_stage, two, blockDim.x);
    if (two) {
        raw35_stage<U>(t, kB, vB, s_stage);
        raw35_finish(a, vrows, t, f, c, kB, s_stage, false, blockDim.x);
    }
    }
}

