// Device baseline-JPEG decoder (frame ingest; the reference decodes every frame with cv2.imread =
// libjpeg(-turbo) on one host core, cama/reproject.py:224,243).  Included by cama_jpeg.hip inside its anonymous namespace.
//
// Bit-exact restatement of libjpeg-turbo's default decode path (JDCT_ISLOW integer IDCT, "fancy" triangle chroma
// upsampling, 16-bit fixed-point YCbCr->RGB), pinned by oracle/jpeg_oracle.py == Pillow's libjpeg-turbo.
//
// Stages, all images of a batch per launch:
//   k_jpeg_count / k_jpeg_tilescan / k_jpeg_unstuff   remove 0xFF00 byte stuffing (count, per-image scan, compact); the scan
//                    also clears the image's status word and the compaction the slack behind the unstuffed bytes -- the chain
//                    has ONE fill in front of it (the coefficients: only non-zero ones are stored)
//   k_jpeg_sync<1>   Huffman decode of every 1024-bit subsequence from a SPECULATIVE start (its first bit, block 0,
//                    DC expected), then a fixpoint inside each 256-subsequence workgroup: a subsequence is re-decoded
//                    from its predecessor's end state until no end state changes.  Huffman streams self-synchronise,
//                    so a wrong start usually reaches the right end state within one or two subsequences.
//   k_jpeg_sync<2>   the same fixpoint seeded with the true entry state of each workgroup (= the previous
//                    workgroup's last end state), which repairs the leading subsequences of every workgroup
//   k_jpeg_write     final decode of every subsequence from its now exact start state; coefficients are written at
//                    block index = exclusive scan of the per-subsequence completed-block counts; every end state is
//                    re-checked, an inconsistency raises the image's status flag (the caller falls back to the host)
//   k_jpeg_dc        per component prefix sum of the DC differences
//   k_jpeg_idct      dequantise + islow IDCT, 8 lanes per block, column pass -> LDS -> row pass -> component planes
//   k_jpeg_colour    fancy upsampling + colour conversion, one thread per output pixel, BGR or RGB
//   k_jpeg_find_restarts   (DRI files, before the descriptors exist) positions of every RSTn marker in the uploaded bytes

#ifndef JPEG_SUB_SHIFT
#define JPEG_SUB_SHIFT 5
#endif
constexpr int JPEG_SUB_WORDS = 1 << JPEG_SUB_SHIFT; // subsequence: 32 dwords = 1024 bits
constexpr int JPEG_SUB_BITS = JPEG_SUB_WORDS * 32;
#ifndef JPEG_WG_N
#define JPEG_WG_N 256
#endif
constexpr int JPEG_WG = JPEG_WG_N;                 // subsequences (threads) per workgroup
constexpr int JPEG_TILE = 1024;                    // unstuff tile, bytes (4 per thread)
#ifndef JPEG_SPEC_SKIP
#define JPEG_SPEC_SKIP 0
#endif
constexpr int JPEG_LUT_BITS = 10;
constexpr int JPEG_L2_MAX = 1024;                  // second-level entries per table set (the standard tables need 674)
constexpr uint32_t JPEG_L2_NONE = 0xffffu;

// Huffman table set as uploaded by the host (cama_amd/jpeg.py: build_huff_set) and copied verbatim to LDS.
// Tables: 0 = DC0, 1 = AC0, 2 = DC1, 3 = AC1.
struct JpegHuffSet {
    uint16_t lut[4][1 << JPEG_LUT_BITS];   // (length << 8) | symbol for codes of <= 10 bits, 0 otherwise
    uint32_t lim[4][8];                    // [t][i], i = 0..5: 16-bit left-aligned exclusive upper limit of the codes
                                           // of length <= 11+i; a longer code's length is 11 + #limits <= its prefix
    int32_t valoff[4][17];                 // index of a length's first symbol minus its first code
    uint8_t vals[4][256];
    // second level for the codes of 11..16 bits (round 5): canonical codes grow with their length, so they occupy the 16-bit
    // prefixes from l2_first[t] << 6 up to the end of the table's code space, top = min(lim[t][5], 0xffff).
    // l2[l2_off[t] + min(prefix16, top) - (l2_first[t] << 6)] = (length << 8) | symbol, the entry at `top` = 16 << 8 (no
    // code) -- exactly what jpeg_symbol_long_t computes, one lookup instead of a compare chain and two dependent reads.
    // l2_off[t] == JPEG_L2_NONE: the set's long codes did not fit (slow path).
    uint16_t l2_first[4];                  // 10-bit prefix of the first lut entry that is 0 (1024: none)
    uint16_t l2_off[4];
    uint16_t l2[JPEG_L2_MAX];
};
static_assert(sizeof(JpegHuffSet) == 9632 + 2 * JPEG_L2_MAX && sizeof(JpegHuffSet) % 16 == 0, "JpegHuffSet layout");

// What the host uploads per table set (cama_jpeg_huff_set_bytes()): the symbol tables, then the TRANSITION tables of the
// synchronisation phases (jpeg_sync_span) in the order k_jpeg_sync keeps them in LDS -- one 14 KB block copy per workgroup.
// Until round 5 every workgroup of k_jpeg_sync<1|2> derived them from the symbol tables itself (two dependent global reads per
// entry, 19 us of a 230 us workgroup and of the 75-130 us ones of phase 2); they depend on the table set alone.
struct JpegSyncSet {
    uint16_t dc[2][1 << JPEG_LUT_BITS];    // used | kinc << 6
    uint32_t ac[2][1 << JPEG_LUT_BITS];    // used1 | kinc1 << 6 | used12 << 13 | kinc12 << 19 (0: code longer than 10 bits)
    uint16_t l2[JPEG_L2_MAX];              // transition entries of the codes of 11..16 bits (JpegHuffSet::l2)
};
struct JpegHuffRec {
    JpegHuffSet set;
    JpegSyncSet sync;
};
static_assert(sizeof(JpegSyncSet) == 14336 && sizeof(JpegHuffRec) == sizeof(JpegHuffSet) + sizeof(JpegSyncSet) &&
              offsetof(JpegHuffRec, sync) % 16 == 0, "JpegHuffRec layout");

struct JpegArgs {
    const uint8_t *stream;            // stuffed entropy segments
    uint8_t *clean;                   // unstuffed copy (same offsets)
    const cama_jpeg_image *imgs;      // device copy of the planned descriptors
    int n;
    const JpegHuffRec *huff;
    const uint16_t *quant;            // [sets][3][64] natural order
    uint32_t *tile_count, *tile_base; // per unstuff tile
    uint32_t *nbits;                  // per image: bits in the unstuffed stream
    uint64_t *E;                      // per subsequence: end state  pos << 16 | blk << 8 | k
    uint32_t *nb;                     // per subsequence: blocks completed
    uint32_t *wg_total;               // per decode workgroup: blocks completed
    int16_t *coef;
    int16_t *dcd;                     // per block (scan order, coef_off / 64): the DC difference, after k_jpeg_dc the DC term itself
    uint8_t *planes;
    uint8_t *out;                     // [n, height, width, 3]
    size_t out_stride;                // bytes between images
    int bgr;
    int32_t *status;                  // per image, 0 = ok
};

__device__ __forceinline__ int jpeg_find_image(const cama_jpeg_image *imgs, int n, uint32_t id, bool by_tile)
{
    int lo = 0, hi = n - 1;           // last image whose first workgroup / tile is <= id
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        const uint32_t first = by_tile ? imgs[mid].tile0 : imgs[mid].wg0;
        if (first <= id) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------ byte unstuffing
// byte i of an entropy segment is dropped iff it is the 0x00 that follows a 0xFF (T.81 B.1.1.5)
__device__ __forceinline__ uint32_t jpeg_drop_mask(const uint8_t *seg, uint32_t len, uint32_t i0, uint32_t &bytes)
{
    // bytes i0..i0+3 (zero beyond len) and which of them are stuffed zeros; i0 < len, i0 % 4 == 0.  The segment starts at
    // any byte: the five bytes i0-1..i0+3 come out of (at most) three ALIGNED dword loads and a byte-align (five byte
    // loads per thread made the two unstuffing passes run at 200 GB/s)
    const uint8_t *p = seg + i0;
    const uint32_t a0 = (uint32_t)(uintptr_t)p & 3u;
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p - a0);
    const uint32_t d1 = q[0];
    const uint32_t d2 = (a0 && i0 + (4u - a0) < len) ? q[1] : 0u;
    bytes = __builtin_amdgcn_alignbyte(d2, d1, a0);
    uint32_t prev = 0;
    if (i0) prev = a0 ? (d1 >> (8u * (a0 - 1u))) & 255u : q[-1] >> 24;
    const uint32_t valid = min(len - i0, 4u);
    if (valid < 4u) bytes &= (1u << (8u * valid)) - 1u;
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t b = (bytes >> (8 * k)) & 255u;
        if ((uint32_t)k < valid && b == 0u && prev == 0xFFu) m |= 1u << k;
        prev = b;
    }
    return m;
}

__global__ __launch_bounds__(JPEG_TILE / 4) void k_jpeg_count(JpegArgs a)
{
    const int img = jpeg_find_image(a.imgs, a.n, blockIdx.x, true);
    const cama_jpeg_image &D = a.imgs[img];
    const uint32_t i0 = (blockIdx.x - D.tile0) * JPEG_TILE + threadIdx.x * 4;
    uint32_t bytes;
    const uint32_t m = i0 < D.stream_len ? jpeg_drop_mask(a.stream + D.stream_off, D.stream_len, i0, bytes) : 0u;
    uint32_t c = __popc(m);
#pragma unroll
    for (int off = 32; off; off >>= 1) c += __shfl_xor(c, off, 64);
    __shared__ uint32_t s_w[JPEG_TILE / 4 / 64];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < JPEG_TILE / 4 / 64; ++w) t += s_w[w];
        a.tile_count[blockIdx.x] = t;
    }
}

// one workgroup per image: exclusive scan of its tiles' drop counts, and the image's unstuffed bit count
__global__ __launch_bounds__(256) void k_jpeg_tilescan(JpegArgs a)
{
    const cama_jpeg_image &D = a.imgs[blockIdx.x];
    __shared__ uint32_t s_part[256];
    const uint32_t per = (D.ntile + 255u) / 256u;
    const uint32_t t0 = min(threadIdx.x * per, D.ntile), t1 = min(t0 + per, D.ntile);
    uint32_t sum = 0;
    for (uint32_t t = t0; t < t1; ++t) sum += a.tile_count[D.tile0 + t];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int k = 0; k < 256; ++k) {
            const uint32_t v = s_part[k];
            s_part[k] = run;
            run += v;
        }
        a.nbits[blockIdx.x] = (D.stream_len - run) * 8u;
        a.status[blockIdx.x] = 0;                 // (k_jpeg_write raises it; no separate fill on the chain)
    }
    __syncthreads();
    uint32_t run = s_part[threadIdx.x];
    for (uint32_t t = t0; t < t1; ++t) {
        a.tile_base[D.tile0 + t] = run;
        run += a.tile_count[D.tile0 + t];
    }
}

__global__ __launch_bounds__(JPEG_TILE / 4) void k_jpeg_unstuff(JpegArgs a)
{
    const int img = jpeg_find_image(a.imgs, a.n, blockIdx.x, true);
    const cama_jpeg_image &D = a.imgs[img];
    const uint32_t i0 = (blockIdx.x - D.tile0) * JPEG_TILE + threadIdx.x * 4;
    uint32_t bytes = 0;
    const uint32_t m = i0 < D.stream_len ? jpeg_drop_mask(a.stream + D.stream_off, D.stream_len, i0, bytes) : 0u;
    // exclusive prefix of the drop counts inside the tile
    const uint32_t c = __popc(m);
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += v;
    }
    __shared__ uint32_t s_w[JPEG_TILE / 4 / 64];
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = a.tile_base[blockIdx.x] + incl - c;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += s_w[w];
    uint8_t *dst = a.clean + D.clean_off;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = i0 + k;
        if (i < D.stream_len) {
            if (m & (1u << k)) ++before;
            else dst[i - before] = (uint8_t)(bytes >> (8 * k));
        }
    }
    // The decoders read up to 35 bytes past a segment and expect zeros behind its last unstuffed byte: the image's last tile
    // clears [unstuffed length, end of the segment's slot) -- as many bytes as were dropped + the slack, instead of a fill of
    // the whole buffer in front of every group's chain (no other tile writes at or past the unstuffed length).
    if (blockIdx.x - D.tile0 == D.ntile - 1u) {
        const uint32_t from = a.nbits[img] >> 3, to = (D.stream_len + 64u + 15u) & ~15u;
        for (uint32_t i = from + threadIdx.x; i < to; i += JPEG_TILE / 4) dst[i] = 0;
    }
}

// ------------------------------------------------------------------------------------------ Huffman decoding
struct JpegState { uint32_t pos, blk, k; };
struct JpegWgCtx;
__device__ __forceinline__ uint32_t jpeg_word(const JpegWgCtx &c, uint32_t w);   // word w (relative to c.word0), big-endian numeric
__device__ __forceinline__ uint64_t jpeg_pack(const JpegState &s) { return ((uint64_t)s.pos << 16) | (s.blk << 8) | s.k; }
__device__ __forceinline__ JpegState jpeg_unpack(uint64_t v)
{
    return JpegState{(uint32_t)(v >> 16), (uint32_t)(v >> 8) & 0xffu, (uint32_t)v & 0xffu};
}

// -DJPEG_WORDS_GLOBAL (A/B switch, off): the decoders read the unstuffed stream straight from global memory (one dword
// per refill) instead of a 34 KB LDS copy per workgroup: 13 KB instead of 47 KB of LDS, 8 instead of 3 workgroups per
// CU, so that the groups of a batch (separate streams) could run side by side.  Measured twice (round 1: one group,
// -3 % ... +2 %; round 2: 240 photo-like frames over 4 and 8 streams): 27-30 k images/s against 36 k with the LDS copy
// -- the refill read sits on the symbol chain's critical path, and LDS latency beats an L1 hit.
struct JpegWgCtx {
    const uint32_t *words;     // LDS: this workgroup's stream words, big-endian numeric, one pad word per 32 (or, with
                               // JPEG_WORDS_GLOBAL, the image's unstuffed segment in global memory, little-endian dwords)
    uint32_t word0;            // image-relative index of words[0] (0 with JPEG_WORDS_GLOBAL)
    uint32_t nwords;           // JPEG_WORDS_GLOBAL: readable dwords of the segment (zero beyond)
    const JpegHuffSet *H;      // LDS (k_jpeg_write)
    // k_jpeg_sync only (JpegSyncShared): transition tables instead of H
    const uint16_t *sync_dc;   // [2][1024]: used | kinc << 6
    const uint32_t *sync_ac;   // [2][1024]: used1 | kinc1 << 6 | used12 << 13 | kinc12 << 19 (0: code longer than 10 bits)
    const JpegHuffSet *G;      // global: the table set (slow path of a set whose long codes have no second level)
    const uint16_t *l2;        // LDS: second level, k_jpeg_write: (length << 8) | symbol, k_jpeg_sync: used | kinc << 6
    // top << 16 | adj per table (0 = DC0, 1 = AC0, 2 = DC1, 3 = AC1): table t's entry for 16-bit prefix p is l2[min(p, top) - adj];
    // JPEG_L2_ABSENT: slow path.  Four SCALARS, not an array: picked by lane-varying selectors, an array was placed in scratch
    // memory and every code longer than 10 bits -- some lane of a wave has one nearly every step -- paid a scratch load
    // (s_waitcnt vmcnt(0)) in front of its second-level lookup.
    uint32_t adj0, adj1, adj2, adj3;
    uint32_t dc_mask, ac_mask; // bit b: Huffman table selector (0/1) of block b of the MCU
    uint32_t bpm;
    const uint8_t *zigzag;     // LDS copy of the zigzag -> natural order table
};
__device__ __forceinline__ uint32_t jpeg_word(const JpegWgCtx &c, uint32_t w)
{
#ifdef JPEG_WORDS_GLOBAL
    return w < c.nwords ? __builtin_bswap32(c.words[w]) : 0u;
#else
    return c.words[w + (w >> JPEG_SUB_SHIFT)];
#endif
}


// codes longer than 10 bits: canonical codes grow with their length, so the length is 11 + the number of per-length
// limits (left-aligned to 16 bits, monotone) that the 16-bit prefix has reached -- no dependent loop
__device__ __forceinline__ uint32_t jpeg_symbol_long_t(const uint32_t (*lim)[8], const int32_t (*valoff)[17],
                                                       const uint8_t (*vals)[256], uint32_t tab, uint32_t window)
{
    const uint32_t w16 = window >> 16;
    const uint4 lim0 = *reinterpret_cast<const uint4 *>(&lim[tab][0]);
    const uint2 lim1 = *reinterpret_cast<const uint2 *>(&lim[tab][4]);
    const uint32_t l = 11u + (w16 >= lim0.x) + (w16 >= lim0.y) + (w16 >= lim0.z) + (w16 >= lim0.w) + (w16 >= lim1.x);
    if (w16 >= lim1.y) return 16u << 8;        // no such code (only on speculative paths): skip 16 bits, symbol 0
    return (l << 8) | vals[tab][((w16 >> (16u - l)) + (uint32_t)valoff[tab][l]) & 255u];
}
__device__ __forceinline__ uint32_t jpeg_symbol_long(const JpegHuffSet &H, uint32_t tab, uint32_t window)
{
    return jpeg_symbol_long_t(H.lim, H.valoff, H.vals, tab, window);
}
constexpr uint32_t JPEG_L2_ABSENT = 0xffffffffu;
// per-table clamp and rebasing of the second level (wave-uniform: a scalar register each), top << 16 | adj: l2[min(prefix16, top) - adj]
__device__ __forceinline__ uint32_t jpeg_l2_adjust1(const JpegHuffSet &G, int t)
{
    const uint32_t top = min(G.lim[t][5], 0xffffu), base = (uint32_t)G.l2_first[t] << 6, off = G.l2_off[t];
    uint32_t v;
    if (G.lim[t][5] == 0u || G.l2_first[t] >= (1u << JPEG_LUT_BITS))
        v = 0u;                        // a table the file does not define (grayscale: tables 2 and 3), or one without long codes:
                                       // its second level is never USED, but every lane of a wave-mode walk evaluates all four
                                       // tables -- top = 0, adj = 0 makes that the in-bounds read l2[0]
    else if (off == JPEG_L2_NONE || base < off || base - off > 0xffffu)
        v = JPEG_L2_ABSENT;            // no second level, or one the 16-bit rebasing cannot express (a table of long codes only)
    else
        v = (top << 16) | (base - off);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// The four values as scalar-register VALUES (read once per function, outside its loop): a lane-varying choice between struct
// fields is turned into an indexed load by the compiler, which then keeps the whole context in scratch memory.
struct JpegAdj { uint32_t a0, a1, a2, a3; };
__device__ __forceinline__ JpegAdj jpeg_adj(const JpegWgCtx &c)
{
    return JpegAdj{(uint32_t)__builtin_amdgcn_readfirstlane((int)c.adj0), (uint32_t)__builtin_amdgcn_readfirstlane((int)c.adj1),
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)c.adj2), (uint32_t)__builtin_amdgcn_readfirstlane((int)c.adj3)};
}
__device__ __forceinline__ uint32_t jpeg_adj_of(const JpegAdj &A, uint32_t tab)      // tab: 0 = DC0, 1 = AC0, 2 = DC1, 3 = AC1
{
    const uint32_t lo = (tab & 1u) ? A.a1 : A.a0, hi = (tab & 1u) ? A.a3 : A.a2;
    return (tab & 2u) ? hi : lo;
}
__device__ __forceinline__ uint32_t jpeg_l2_index(uint32_t adj, uint32_t window)
{
    // (a prefix below the table's first long code -- a lane that evaluates a table its symbol does not use -- clamps to entry 0)
    return (uint32_t)max((int32_t)min(window >> 16, adj >> 16) - (int32_t)(adj & 0xffffu), 0);
}
// a code longer than 10 bits, in k_jpeg_write's form (length << 8 | symbol)
__device__ __forceinline__ uint32_t jpeg_long_symbol(const JpegWgCtx &c, const JpegAdj &A, uint32_t tab, uint32_t window)
{
    const uint32_t adj = jpeg_adj_of(A, tab);
    if (adj != JPEG_L2_ABSENT) return c.l2[jpeg_l2_index(adj, window)];
    return jpeg_symbol_long(*c.G, tab, window);
}
// the same in k_jpeg_sync's form (used | kinc << 6): its second level holds transition entries
__device__ __forceinline__ uint32_t jpeg_sync_entry(uint32_t tab, uint32_t e);
__device__ __forceinline__ uint32_t jpeg_long_entry(const JpegWgCtx &c, const JpegAdj &A, uint32_t tab, uint32_t window)
{
    const uint32_t adj = jpeg_adj_of(A, tab);
    if (adj != JPEG_L2_ABSENT) return c.l2[jpeg_l2_index(adj, window)];
    return jpeg_sync_entry(tab & 1u, jpeg_symbol_long(*c.G, tab, window));
}

__constant__ uint8_t c_jpeg_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                          12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                          58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Decode every symbol that STARTS in [s.pos, end): T.81 F.2.2 with the decoder state (block in MCU, zigzag index).
// A lone wave issues one instruction every ~11 cycles here (measured), so the loop is written for instruction count:
// the bit window is a 64-bit value in two 32-bit registers (hi always holds 32 valid bits; v_alignbit shifts, one
// unconditional LDS read per symbol for the refill word), table choice and state update are selects, and the only
// divergent branches are the >10-bit codes and the coefficient store.  WRITE: store the coefficients of block
// `block` onwards (DC as the raw difference).
template <bool WRITE>
__device__ __forceinline__ uint32_t jpeg_decode_span(const JpegWgCtx &c, JpegState &s, uint32_t end, int16_t *coef,
                                                     int16_t *dcd, uint32_t block, uint32_t total_blocks)
{
    if (s.pos >= end) return 0;
    uint32_t nb = 0, pos = s.pos, blk = s.blk, k = s.k;
    uint32_t w = (pos >> 5) - c.word0;
    uint32_t hi, lo, cnt;                                           // cnt = valid bits in lo (all of hi is valid)
    {
        const uint32_t w0 = jpeg_word(c, w), w1 = jpeg_word(c, w + 1), sh = pos & 31u;
        hi = sh ? __builtin_amdgcn_alignbit(w0, w1, 32u - sh) : w0;
        lo = sh ? w1 << sh : w1;
        cnt = 32u - sh;
        w += 2;
    }
    const uint16_t *lut = &c.H->lut[0][0];
    const JpegAdj A = jpeg_adj(c);
    while (pos < end) {
        const uint32_t isac = k ? 1u : 0u;
        const uint32_t tab = (((isac ? c.ac_mask : c.dc_mask) >> blk) & 1u) * 2u + isac;
        uint32_t e = lut[(tab << JPEG_LUT_BITS) + (hi >> (32 - JPEG_LUT_BITS))];
        const uint32_t next = jpeg_word(c, w);                     // refill word (used when lo runs dry)
        if (e == 0u) e = jpeg_long_symbol(c, A, tab, hi);   // (read beside the others as in jpeg_sync_span: measured, no gain here)
        const uint32_t len = e >> 8, size = e & 15u, run = (e >> 4) & 15u;
        // zigzag index of the coded coefficient (DC: 0) and the index after this symbol
        const uint32_t at = isac ? k + run : 0u;
        const uint32_t knext = isac ? (size ? at + 1u : (run == 15u ? k + 16u : 64u)) : 1u;
        if (WRITE && (size | (isac ^ 1u)) && at < 64u && block + nb < total_blocks) {
            // (a DC symbol stores its difference even when that is zero: the compact array needs no fill in front of the chain)
            const uint32_t sz = size ? size : 1u;
            const uint32_t v = (hi << len) >> (32u - sz);
            int32_t val = (v < (1u << (sz - 1u))) ? (int32_t)v - (int32_t)((1u << sz) - 1u) : (int32_t)v;
            val = size ? val : 0;
            // the DC difference goes to its own compact array: the prefix sum over a component's blocks (k_jpeg_dc) walked the
            // coefficient buffer with a 128-byte stride otherwise, every cache line of it once more (640 of a 4 500 us batch)
            if (isac) coef[(size_t)(block + nb) * 64 + c.zigzag[at]] = (int16_t)val;
            else dcd[block + nb] = (int16_t)val;
        }
        const uint32_t used = len + size;                            // 1..31
        pos += used;
        hi = __builtin_amdgcn_alignbit(hi, lo, 32u - used);
        lo <<= used;
        // lo holds cnt valid bits at its top; after consuming `used`: cnt - used (may go negative -> refill)
        const int32_t left = (int32_t)cnt - (int32_t)used;
        if (left < 0) {
            // hi is short of -left bits at its bottom: they are the top bits of `next`
            const uint32_t miss = (uint32_t)(-left);
            hi |= next >> (32u - miss);
            lo = next << miss;
            cnt = 32u - miss;
            ++w;
        } else {
            cnt = (uint32_t)left;
        }
        const bool done = knext >= 64u;
        k = done ? 0u : knext;
        blk = done ? (blk + 1u == c.bpm ? 0u : blk + 1u) : blk;
        nb += done ? 1u : 0u;
    }
    s.pos = pos;
    s.blk = blk;
    s.k = k;
    return nb;
}

// ---- the synchronisation phases' own decoder (round 5) --------------------------------------------------------------
// k_jpeg_sync<1|2> never need a coefficient's VALUE: only how many bits a symbol takes and where it leaves the zigzag index.
// Their workgroups therefore keep TRANSITION tables in LDS instead of the symbol LUT (JpegSyncSet, built by the host):
//     used = code length + magnitude bits (6 bits)       kinc: the zigzag index after the symbol is k + kinc (7 bits;
//                                                               DC: 1; AC with a value: run + 1; ZRL: 16; EOB: 64)
// -- the same state transitions as jpeg_decode_span (T.81 F.2.2), with "k + kinc >= 64" as the one end-of-block test: ~30
// instructions per symbol with two rare branches where jpeg_decode_span<false> compiled to ~58 with five divergent regions.
// The chain is latency-bound, though (one dependent LDS lookup per symbol at 3 waves per SIMD: the leaner loop alone gave
// +5 %), so an AC entry also carries the symbol AFTER it when that one's code lies inside the 10 known bits too:
//     used1 | kinc1 << 6 | used12 << 13 | kinc12 << 19          (used12 = used1, kinc12 = kinc1: no second symbol)
// and one lookup advances two symbols whenever the first does not end the block and the second still starts inside the
// subsequence -- exactly what two single steps would have done (same table: the block, hence the selector, is unchanged;
// same code: its bits are all known).  The phases are 62.7 of a 79 ms decode pass (profiles/r04_demo_loop_timeline.txt).
__device__ __forceinline__ uint32_t jpeg_sync_entry(uint32_t tab, uint32_t e)       // e = (length << 8) | symbol, != 0
{
    const uint32_t len = e >> 8, size = e & 15u, run = (e >> 4) & 15u;
    const uint32_t kinc = (tab & 1u) ? (size ? run + 1u : (run == 15u ? 16u : 64u)) : 1u;
    return (len + size) | (kinc << 6);
}

// every symbol that STARTS in [s.pos, end), state only; c.sync_dc / c.sync_ac hold the transition tables
__device__ __forceinline__ uint32_t jpeg_sync_span(const JpegWgCtx &c, JpegState &s, uint32_t end)
{
    if (s.pos >= end) return 0;
    uint32_t nb = 0, pos = s.pos, blk = s.blk, k = s.k;
    uint32_t w = (pos >> 5) - c.word0;
    uint32_t hi, lo;
    int32_t cnt;                                                    // valid bits in lo (all of hi is valid)
    {
        const uint32_t w0 = jpeg_word(c, w), w1 = jpeg_word(c, w + 1), sh = pos & 31u;
        hi = sh ? __builtin_amdgcn_alignbit(w0, w1, 32u - sh) : w0;
        lo = sh ? w1 << sh : w1;
        cnt = 32 - (int32_t)sh;
        w += 2;
    }
    const uint32_t bpm = c.bpm;
    const JpegAdj A = jpeg_adj(c);
    while (pos < end) {
        // One round of FOUR independent LDS reads per step -- the DC entry, the AC entry, the second-level entry (as if the
        // code were longer than 10 bits) and the refill word -- then selects: the lanes of a wave are at different places of
        // their blocks, so a branch per case (DC / AC / long code / refill) was taken by SOME lane nearly every step and the
        // wave paid every body plus three dependent LDS latencies (62 -> 46 us per subsequence with the second level alone).
        const uint32_t peek = hi >> (32 - JPEG_LUT_BITS);
        const uint32_t next = jpeg_word(c, w);                      // refill word (used when lo runs dry)
        const bool isdc = k == 0u;
        const uint32_t sel = ((isdc ? c.dc_mask : c.ac_mask) >> blk) & 1u;
        const uint32_t at = (sel << JPEG_LUT_BITS) + peek;
        const uint32_t edc = c.sync_dc[at], eac = c.sync_ac[at];
        const uint32_t adj = sel ? (isdc ? A.a2 : A.a3) : (isdc ? A.a0 : A.a1);
        const int32_t i2 = (int32_t)min(hi >> 16, adj >> 16) - (int32_t)(adj & 0xffffu);   // (< 0: not a long code; absent: <= 0)
        uint32_t e2 = c.l2[max(i2, 0)];
        asm volatile("" : "+v"(e2));                                 // (keeps the read HERE, beside the other three: the compiler
                                                                    //  sank it into the long-code branch, a dependent LDS latency)
        uint32_t e = isdc ? edc | (edc << 13) : eac;
        if (e == 0u) {
            e = e2 | (e2 << 13);
            if (adj == JPEG_L2_ABSENT) {                                // (a table set without a second level: never the usual files)
                e = jpeg_long_entry(c, A, 2u * sel + (isdc ? 0u : 1u), hi);
                e |= e << 13;
            }
        }
        const uint32_t u1 = e & 63u, k1 = (e >> 6) & 127u;
        // both symbols, unless the first ends the block (the next is then a DC symbol) or the second starts past `end`
        // (a DC entry carries itself twice)
        const bool both = (k + k1 < 64u) && (pos + u1 < end);
        const uint32_t used = both ? (e >> 13) & 63u : u1;
        const uint32_t kinc = both ? e >> 19 : k1;
        k += kinc;
        pos += used;
        hi = __builtin_amdgcn_alignbit(hi, lo, 32u - used);
        lo <<= used;
        cnt -= (int32_t)used;
        if (cnt < 0) {                                              // hi is short of -cnt bits at its bottom: the top bits of `next`
            const uint32_t miss = (uint32_t)(-cnt);
            hi |= next >> (32u - miss);
            lo = next << miss;
            cnt = 32 - (int32_t)miss;
            ++w;
        }
        const bool done = k >= 64u;
        const uint32_t blk1 = blk + 1u == bpm ? 0u : blk + 1u;
        k = done ? 0u : k;
        blk = done ? blk1 : blk;
        nb += done ? 1u : 0u;
    }
    s.pos = pos;
    s.blk = blk;
    s.k = k;
    return nb;
}

// The same decode executed by a whole WAVE for ONE subsequence: lane i looks up, under all four tables, the symbol
// that would start at bit pos + i; the wave then walks the actual symbol chain with scalar state and v_readlane
// (no memory access per symbol).  A lone lane needs ~700 cycles per symbol (dependent LDS lookup + ~75 instructions
// at one instruction per ~11 cycles); this needs one round of lookups per 64 bits plus ~20 scalar instructions per
// symbol.  64x less throughput per wave, ~5x less latency: used where only a few subsequences are left to re-decode,
// i.e. exactly the serial chains that set the kernel time.  State transitions are identical to jpeg_decode_span.
__device__ __forceinline__ uint32_t jpeg_decode_span_wave(const JpegWgCtx &c, JpegState &s, uint32_t end)
{
    const uint32_t lane = threadIdx.x & 63u;
    end = (uint32_t)__builtin_amdgcn_readfirstlane((int)end);
    uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.pos), blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.blk),
             k = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.k), nb = 0;            // wave-uniform
    const JpegAdj A = jpeg_adj(c);
    while (pos < end) {
        const uint32_t bp = pos + lane;
        const uint32_t w = (bp >> 5) - c.word0, sh = bp & 31u;
        const uint32_t hi = jpeg_word(c, w), lo = jpeg_word(c, w + 1);
        const uint32_t win = sh ? __builtin_amdgcn_alignbit(hi, lo, 32u - sh) : hi;
        // single-symbol transition entries (used | kinc << 6) under all four tables: 0 = DC0, 1 = AC0, 2 = DC1, 3 = AC1
        const uint32_t peek = win >> (32 - JPEG_LUT_BITS);
        uint32_t e0 = c.sync_dc[peek], e2 = c.sync_dc[(1u << JPEG_LUT_BITS) + peek];
        uint32_t e1 = c.sync_ac[peek] & 0x1fffu, e3 = c.sync_ac[(1u << JPEG_LUT_BITS) + peek] & 0x1fffu;
        if (e0 == 0u) e0 = jpeg_long_entry(c, A, 0u, win);
        if (e1 == 0u) e1 = jpeg_long_entry(c, A, 1u, win);
        if (e2 == 0u) e2 = jpeg_long_entry(c, A, 2u, win);
        if (e3 == 0u) e3 = jpeg_long_entry(c, A, 3u, win);
        // The walk, in scalar registers: a block's DC symbol, then its AC symbols under ONE table vector (chosen once per block
        // and window).  Bit offset and zigzag index advance in one add -- t = (off + 0x8000 - lim) | (k + 0x8000 - 64) << 16,
        // an AC entry re-packed as used | kinc << 16 -- and "window ran out or block ended" is one mask test (no field can
        // carry: off < 128, k < 128).  Six scalar instructions per AC symbol; a generic per-symbol table choice cost 33, at one
        // instruction per ~10 cycles: 27 us per 1024-bit subsequence.
        const uint32_t lim = min(64u, end - pos);
        const uint32_t a1 = (e1 & 63u) | ((e1 >> 6) << 16), a3 = (e3 & 63u) | ((e3 >> 6) << 16);
        const uint32_t boff = 0x8000u - lim, bk = 0x8000u - 64u;
        uint32_t off = 0;
        for (;;) {
            if (k == 0u) {
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)e0, (int)off);
                const uint32_t d2 = (uint32_t)__builtin_amdgcn_readlane((int)e2, (int)off);
                const uint32_t e = ((c.dc_mask >> blk) & 1u) ? d2 : d0;
                off += e & 63u;
                k = e >> 6;                                              // (1: a DC symbol)
                if (off >= lim) break;
            }
            const uint32_t vac = ((c.ac_mask >> blk) & 1u) ? a3 : a1;    // wave-uniform choice
            uint32_t t = (off + boff) | ((k + bk) << 16);                // here off < lim and k < 64
            do {
                t += (uint32_t)__builtin_amdgcn_readlane((int)vac, (int)((t - boff) & 63u));
            } while ((t & 0x80008000u) == 0u);
            off = (t & 0xffffu) - boff;
            k = (t >> 16) - bk;
            if (k < 64u) break;                                          // the window (or the subsequence) ran out
            k = 0u;
            blk = blk + 1u == c.bpm ? 0u : blk + 1u;
            ++nb;
            if (off >= lim) break;
        }
        pos += off;
    }
    s.pos = pos;
    s.blk = blk;
    s.k = k;
    return nb;
}

// common prologue: stream words + tables to LDS
struct JpegWgShared {
    JpegHuffSet H;
#ifndef JPEG_WORDS_GLOBAL
    uint32_t words[JPEG_WG * JPEG_SUB_WORDS + JPEG_WG + 40];   // one pad word per subsequence: lane stride odd
#endif
    uint64_t E[JPEG_WG];
    uint32_t nb[JPEG_WG];
    uint8_t flag[2][JPEG_WG];
    uint8_t zigzag[64];
};

// the workgroup's share of the unstuffed stream into LDS + the per-image decoder constants
__device__ __forceinline__ void jpeg_wg_stream(const JpegArgs &a, const cama_jpeg_image &D, uint32_t lw, uint32_t *words,
                                               JpegWgCtx &c)
{
    // stream: dwords [lw*256*32, +256*32 + 2) of the unstuffed segment (clean_off is 16-byte aligned; the region
    // past the segment is zero: the clean buffer is cleared per batch)
    const uint32_t *g = reinterpret_cast<const uint32_t *>(a.clean + D.clean_off);
    const uint32_t nwords_img = (D.stream_len + 3u) / 4u + 8u;          // slack words exist (plan pads every segment)
#ifdef JPEG_WORDS_GLOBAL
    (void)lw;
    (void)words;
    c.words = g;
    c.word0 = 0;
    c.nwords = nwords_img;
#else
    const uint32_t w0 = lw * JPEG_WG * JPEG_SUB_WORDS;
    // four words per load (clean_off and w0 are multiples of four words; the slot of a segment is readable 64 bytes past its
    // end, nwords_img ends 35 past it at most): 9 loads per thread instead of 33
    const uint4 *g4 = reinterpret_cast<const uint4 *>(g + w0);
    for (uint32_t i = threadIdx.x * 4u; i < JPEG_WG * JPEG_SUB_WORDS + 8; i += JPEG_WG * 4u) {
        const uint32_t w = w0 + i;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (w < nwords_img) v = g4[i >> 2];
        uint32_t *d = &words[i + (i >> JPEG_SUB_SHIFT)];               // (i % 4 == 0: the four words share one pad offset)
        d[0] = w < nwords_img ? __builtin_bswap32(v.x) : 0u;
        d[1] = w + 1u < nwords_img ? __builtin_bswap32(v.y) : 0u;
        d[2] = w + 2u < nwords_img ? __builtin_bswap32(v.z) : 0u;
        d[3] = w + 3u < nwords_img ? __builtin_bswap32(v.w) : 0u;
    }
    c.words = words;
    c.word0 = w0;
    c.nwords = 0;
#endif
    c.bpm = D.bpm;
    uint32_t b = 0;
    c.dc_mask = c.ac_mask = 0;
    const uint32_t luma_blocks = D.hs * D.vs;
    for (uint32_t ci = 0; ci < D.ncomp; ++ci) {
        const uint32_t nblk = ci == 0 ? luma_blocks : 1u;
        for (uint32_t j = 0; j < nblk; ++j, ++b) {
            c.dc_mask |= (D.comp_dc[ci] & 1u) << b;
            c.ac_mask |= (D.comp_ac[ci] & 1u) << b;
        }
    }
}

__device__ __forceinline__ void jpeg_wg_setup(const JpegArgs &a, const cama_jpeg_image &D, uint32_t lw, JpegWgShared &S,
                                              JpegWgCtx &c)
{
    // tables
    const uint4 *src = reinterpret_cast<const uint4 *>(&a.huff[D.huff_set].set);
    uint4 *dst = reinterpret_cast<uint4 *>(&S.H);
    for (uint32_t i = threadIdx.x; i < sizeof(JpegHuffSet) / 16; i += JPEG_WG) dst[i] = src[i];
#ifndef JPEG_WORDS_GLOBAL
    jpeg_wg_stream(a, D, lw, S.words, c);
#else
    jpeg_wg_stream(a, D, lw, nullptr, c);
#endif
    if (threadIdx.x < 64) S.zigzag[threadIdx.x] = c_jpeg_zigzag[threadIdx.x];
    c.zigzag = S.zigzag;
    c.H = &S.H;
    c.G = &a.huff[D.huff_set].set;
    c.l2 = S.H.l2;
    c.adj0 = jpeg_l2_adjust1(*c.G, 0); c.adj1 = jpeg_l2_adjust1(*c.G, 1); c.adj2 = jpeg_l2_adjust1(*c.G, 2); c.adj3 = jpeg_l2_adjust1(*c.G, 3);
    c.sync_dc = nullptr; c.sync_ac = nullptr;
}

// k_jpeg_sync's LDS: transition tables (JpegSyncSet, copied from the table set's record) instead of the symbol LUT -- 51.9 KB,
// three workgroups per CU
struct JpegSyncShared {
    JpegSyncSet T;
#ifndef JPEG_WORDS_GLOBAL
    uint32_t words[JPEG_WG * JPEG_SUB_WORDS + JPEG_WG + 40];   // one pad word per subsequence: lane stride odd
#endif
    uint64_t E[JPEG_WG];
    uint32_t nb[JPEG_WG];
    uint8_t flag[2][JPEG_WG];
};
static_assert(sizeof(JpegSyncShared) <= 52 * 1024, "three k_jpeg_sync workgroups per CU");

__device__ __forceinline__ void jpeg_sync_setup(const JpegArgs &a, const cama_jpeg_image &D, uint32_t lw, JpegSyncShared &S,
                                                JpegWgCtx &c)
{
    const JpegHuffRec &R = a.huff[D.huff_set];
    const JpegHuffSet &G = R.set;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(&R.sync);
        uint4 *dst = reinterpret_cast<uint4 *>(&S.T);
        for (uint32_t i = threadIdx.x; i < sizeof(JpegSyncSet) / 16; i += JPEG_WG) dst[i] = src[i];
    }
#ifndef JPEG_WORDS_GLOBAL
    jpeg_wg_stream(a, D, lw, S.words, c);
#else
    jpeg_wg_stream(a, D, lw, nullptr, c);
#endif
    c.H = nullptr;
    c.zigzag = nullptr;
    c.sync_dc = &S.T.dc[0][0];
    c.sync_ac = &S.T.ac[0][0];
    c.G = &G;
    c.l2 = S.T.l2;
    c.adj0 = jpeg_l2_adjust1(G, 0); c.adj1 = jpeg_l2_adjust1(G, 1); c.adj2 = jpeg_l2_adjust1(G, 2); c.adj3 = jpeg_l2_adjust1(G, 3);
}


// PHASE 1: speculative decode + fixpoint inside the workgroup.  PHASE 2: fixpoint seeded with the true entry state.
template <int PHASE>
__global__ __launch_bounds__(JPEG_WG) void k_jpeg_sync(JpegArgs a)
{
    __shared__ JpegSyncShared S;
    const int img = jpeg_find_image(a.imgs, a.n, blockIdx.x, false);
    const cama_jpeg_image &D = a.imgs[img];
    const uint32_t lw = blockIdx.x - D.wg0;
    const uint32_t nbits = a.nbits[img];
    const uint32_t sub0 = lw * JPEG_WG;                                  // image-relative first subsequence
    if ((uint64_t)sub0 * JPEG_SUB_BITS >= nbits) {                      // workgroup beyond the unstuffed stream
        if (threadIdx.x == 0) a.wg_total[blockIdx.x] = 0;
        return;
    }
    if (PHASE == 2 && lw == 0) return;                                  // its entry state was exact in phase 1
    JpegWgCtx c;
    jpeg_sync_setup(a, D, lw, S, c);                                    // transition tables (state only) + stream words
    const uint32_t nsub_img = (nbits + JPEG_SUB_BITS - 1) / JPEG_SUB_BITS;
    const uint32_t last = min((uint32_t)JPEG_WG - 1u, nsub_img - 1u - sub0);   // last subsequence with bits
    const uint32_t t = sub0 + threadIdx.x;
    const bool active = threadIdx.x <= last;
    const uint32_t lo = t * JPEG_SUB_BITS, hi = min(lo + JPEG_SUB_BITS, nbits);
    const size_t gsub = (size_t)D.wg0 * JPEG_WG + (size_t)lw * JPEG_WG + threadIdx.x;   // global subsequence slot
    int cur = 0;
    __syncthreads();
    if (PHASE == 1) {
        // (thread 0's decode is never repeated in this phase -- for the image's first workgroup it IS the true start)
        JpegState s{threadIdx.x ? min(lo + (uint32_t)JPEG_SPEC_SKIP, hi) : lo, 0u, 0u};
        uint32_t nb = 0;
        if (active) nb = jpeg_sync_span(c, s, hi);
        S.E[threadIdx.x] = jpeg_pack(s);
        S.nb[threadIdx.x] = nb;
        S.flag[0][threadIdx.x] = 1;
    } else {
        S.E[threadIdx.x] = a.E[gsub];
        S.nb[threadIdx.x] = a.nb[gsub];
        S.flag[0][threadIdx.x] = 0;
        __syncthreads();
        if (threadIdx.x < 64) {
            // entry = end state of the previous workgroup's last subsequence; decoded by wave 0 as a whole
            JpegState s = jpeg_unpack(a.E[(size_t)blockIdx.x * JPEG_WG - 1]);
            const uint32_t hi0 = min((sub0 + 1u) * JPEG_SUB_BITS, nbits);
            const uint32_t nb = jpeg_decode_span_wave(c, s, hi0);
            const uint64_t e = jpeg_pack(s);
            if (threadIdx.x == 0) {
                S.flag[0][0] = (e != S.E[0]) || (nb != S.nb[0]);
                S.E[0] = e;
                S.nb[0] = nb;
            }
        }
    }
    // Fixpoint rounds.  While many subsequences changed, every thread re-decodes its own; once at most
    // JPEG_WAVE_MAX are left (the serial chains), each is re-decoded by a whole wave (jpeg_decode_span_wave).
    // (a thread-mode round costs a whole subsequence's latency, 31-37 us, however few lanes take part; a wave walks one in 12.5:
    // with four waves, eight left = two walks each = 25 us.  4 -> 8: +3 % decoder rate, profiles/r05_jpeg_decoder.txt)
    constexpr uint32_t JPEG_WAVE_MAX = 8;
    __shared__ uint16_t s_list[JPEG_WG];
    __shared__ uint64_t s_start[JPEG_WAVE_MAX];
    __shared__ uint32_t s_wcount[JPEG_WG / 64];
    for (int round = 0; round < JPEG_WG; ++round) {
        __syncthreads();
        const bool redo = active && threadIdx.x > 0 && S.flag[cur][threadIdx.x - 1];
        const uint64_t start = threadIdx.x > 0 ? S.E[threadIdx.x - 1] : 0ull;
        const uint64_t mask = __ballot(redo);
        if ((threadIdx.x & 63) == 0) s_wcount[threadIdx.x >> 6] = (uint32_t)__popcll(mask);
        S.flag[cur ^ 1][threadIdx.x] = 0;
        __syncthreads();                                                 // every start is read before any end is written
        uint32_t before = 0, m = 0;
#pragma unroll
        for (int wv = 0; wv < JPEG_WG / 64; ++wv) {
            const uint32_t cnt = s_wcount[wv];
            before += wv < (int)(threadIdx.x >> 6) ? cnt : 0u;
            m += cnt;
        }
        if (m == 0u) break;                                              // uniform
        if (m > JPEG_WAVE_MAX) {
            if (redo) {
                JpegState st = jpeg_unpack(start);
                const uint32_t nb = jpeg_sync_span(c, st, hi);
                const uint64_t e = jpeg_pack(st);
                S.flag[cur ^ 1][threadIdx.x] = (uint8_t)(e != S.E[threadIdx.x]);
                S.E[threadIdx.x] = e;
                S.nb[threadIdx.x] = nb;  // the count depends on the start state even when the end state does not
            }
        } else {
            if (redo) {
                const uint32_t q = before + (uint32_t)__popcll(mask & ((1ull << (threadIdx.x & 63)) - 1ull));
                s_list[q] = (uint16_t)threadIdx.x;
                s_start[q] = start;
            }
            __syncthreads();
            for (uint32_t q = threadIdx.x >> 6; q < m; q += JPEG_WG / 64) {
                const uint32_t j = s_list[q];
                JpegState st = jpeg_unpack(s_start[q]);
                const uint32_t nb = jpeg_decode_span_wave(c, st, min((sub0 + j + 1u) * JPEG_SUB_BITS, nbits));
                const uint64_t e = jpeg_pack(st);
                if ((threadIdx.x & 63) == 0) {
                    S.flag[cur ^ 1][j] = (uint8_t)(e != S.E[j]);
                    S.E[j] = e;
                    S.nb[j] = nb;
                }
            }
        }
        cur ^= 1;
    }
    __syncthreads();
    a.E[gsub] = S.E[min(threadIdx.x, last)];                              // slots past `last` mirror it (entry of the next wg)
    a.nb[gsub] = active ? S.nb[threadIdx.x] : 0u;
    // workgroup total
    uint32_t v = active ? S.nb[threadIdx.x] : 0u;
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off, 64);
    __shared__ uint32_t s_tot[JPEG_WG / 64];
    if ((threadIdx.x & 63) == 0) s_tot[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < JPEG_WG / 64; ++w) tot += s_tot[w];
        a.wg_total[blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(JPEG_WG) void k_jpeg_write(JpegArgs a)
{
    __shared__ JpegWgShared S;
    const int img = jpeg_find_image(a.imgs, a.n, blockIdx.x, false);
    const cama_jpeg_image &D = a.imgs[img];
    const uint32_t lw = blockIdx.x - D.wg0;
    const uint32_t nbits = a.nbits[img];
    const uint32_t sub0 = lw * JPEG_WG;
    if ((uint64_t)sub0 * JPEG_SUB_BITS >= nbits) return;
    JpegWgCtx c;
    jpeg_wg_setup(a, D, lw, S, c);
    const uint32_t nsub_img = (nbits + JPEG_SUB_BITS - 1) / JPEG_SUB_BITS;
    const uint32_t last = min((uint32_t)JPEG_WG - 1u, nsub_img - 1u - sub0);
    const uint32_t t = sub0 + threadIdx.x;
    const bool active = threadIdx.x <= last;
    const uint32_t lo = t * JPEG_SUB_BITS, hi = min(lo + JPEG_SUB_BITS, nbits);
    const size_t gsub = (size_t)D.wg0 * JPEG_WG + (size_t)lw * JPEG_WG + threadIdx.x;
    // blocks completed before this workgroup, then before this subsequence: a reduction and an exclusive scan over the
    // workgroup, by wave shuffles + one LDS hop across the waves (thread 0 walking 256 LDS entries twice cost ~20 us of
    // latency per workgroup)
    uint32_t before = 0;
    for (uint32_t w = threadIdx.x; w < lw; w += JPEG_WG) before += a.wg_total[D.wg0 + w];
    const uint32_t mine = active ? a.nb[gsub] : 0u;
    uint32_t red = before, incl = mine;
#pragma unroll
    for (int off = 32; off; off >>= 1) red += __shfl_xor(red, off, 64);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += v;
    }
    __shared__ uint32_t s_red[JPEG_WG / 64], s_inc[JPEG_WG / 64];
    if ((threadIdx.x & 63) == 63) {
        s_red[threadIdx.x >> 6] = red;
        s_inc[threadIdx.x >> 6] = incl;
    }
    __syncthreads();                                    // (also: the stream words and tables of jpeg_wg_setup)
    uint32_t first_block = incl - mine;                 // exclusive prefix inside the wave ...
#pragma unroll
    for (int w = 0; w < JPEG_WG / 64; ++w) {
        first_block += s_red[w];                         // ... + everything before the workgroup
        if (w < (int)(threadIdx.x >> 6)) first_block += s_inc[w];   // ... + the waves before this one
    }
    if (!active) return;
    JpegState s = (t == 0) ? JpegState{0u, 0u, 0u} : jpeg_unpack(a.E[gsub - 1]);
    const uint32_t nb = jpeg_decode_span<true>(c, s, hi, a.coef + D.coef_off, a.dcd + D.coef_off / 64, first_block, D.total_blocks);
    int bad = (jpeg_pack(s) != a.E[gsub]) || (nb != mine);
    if (t == nsub_img - 1u && first_block + nb != D.total_blocks) bad |= 2;
    if (bad) atomicOr(&a.status[img], bad);
}

// per (image, component): DC[i] = sum of the differences up to block i of that component (T.81 F.2.1.3.1).
// One workgroup per component over the compact array of DC differences (k_jpeg_write: dcd, one int16 per block in scan
// order); each thread gathers its consecutive blocks in batches of independent loads.
constexpr int JPEG_DC_THREADS = 1024;
constexpr int JPEG_DC_BATCH = 8;
__global__ __launch_bounds__(JPEG_DC_THREADS) void k_jpeg_dc(JpegArgs a)
{
    const cama_jpeg_image &D = a.imgs[blockIdx.x];
    const uint32_t ci = blockIdx.y;
    if (ci >= D.ncomp || D.kind == CAMA_JPEG_PIXELS) return;     // restart intervals: one scan per interval (DC resets)
    const uint32_t hv = ci == 0 ? D.hs * D.vs : 1u;
    const uint32_t first = ci == 0 ? 0u : D.hs * D.vs + (ci - 1u);
    const uint32_t n = D.mx * D.my * hv;
    int16_t *dcd = a.dcd + D.coef_off / 64;
    const uint32_t per = (n + JPEG_DC_THREADS - 1u) / JPEG_DC_THREADS;
    const uint32_t i0 = min(threadIdx.x * per, n), i1 = min(i0 + per, n);
    // scan-order block of this component's i-th block: (i / hv) * bpm + first + i % hv, advanced incrementally
    const uint32_t mcu0 = i0 / hv, sub0 = i0 - mcu0 * hv;
    int32_t sum = 0;
    {
        uint32_t mcu = mcu0, sub = sub0;
        for (uint32_t i = i0; i < i1; i += JPEG_DC_BATCH) {
            int32_t v[JPEG_DC_BATCH];
#pragma unroll
            for (int q = 0; q < JPEG_DC_BATCH; ++q) {
                v[q] = (i + q < i1) ? (int32_t)dcd[mcu * D.bpm + first + sub] : 0;
                if (++sub == hv) { sub = 0; ++mcu; }
            }
#pragma unroll
            for (int q = 0; q < JPEG_DC_BATCH; ++q) sum += v[q];
        }
    }
    // exclusive scan of the per-thread sums: inside each wave by shuffles, across the 16 waves through LDS
    int32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t v = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += v;
    }
    __shared__ int32_t s_wave[JPEG_DC_THREADS / 64];
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    int32_t run = incl - sum;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) run += s_wave[w];
    uint32_t mcu = mcu0, sub = sub0;
    for (uint32_t i = i0; i < i1; i += JPEG_DC_BATCH) {
        int32_t v[JPEG_DC_BATCH];
        int16_t *ptr[JPEG_DC_BATCH];
#pragma unroll
        for (int q = 0; q < JPEG_DC_BATCH; ++q) {
            ptr[q] = dcd + (mcu * D.bpm + first + sub);
            v[q] = (i + q < i1) ? (int32_t)*ptr[q] : 0;
            if (++sub == hv) { sub = 0; ++mcu; }
        }
#pragma unroll
        for (int q = 0; q < JPEG_DC_BATCH; ++q) {
            run += v[q];
            if (i + q < i1) *ptr[q] = (int16_t)run;
        }
    }
}

// ------------------------------------------------------------------------------------------ IDCT (IJG jidctint "islow")
__device__ __forceinline__ void jpeg_idct8(const int32_t d[8], int32_t o[8], int shift)
{
    constexpr int32_t F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
                      F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995,
                      F_3_072 = 25172;
    int32_t z1 = (d[2] + d[6]) * F_0_541;
    const int32_t tmp2 = z1 + d[6] * (-F_1_847), tmp3 = z1 + d[2] * F_0_765;
    const int32_t tmp0 = (int32_t)((uint32_t)(d[0] + d[4]) << 13), tmp1 = (int32_t)((uint32_t)(d[0] - d[4]) << 13);
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    int32_t t0 = d[7], t1 = d[5], t2 = d[3], t3 = d[1];
    z1 = t0 + t3;
    int32_t z2 = t1 + t2, z3 = t0 + t2, z4 = t1 + t3;
    const int32_t z5 = (z3 + z4) * F_1_175;
    t0 *= F_0_298; t1 *= F_2_053; t2 *= F_3_072; t3 *= F_1_501;
    z1 *= -F_0_899; z2 *= -F_2_562;
    z3 = z3 * (-F_1_961) + z5;
    z4 = z4 * (-F_0_390) + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    const int32_t r = 1 << (shift - 1);
    o[0] = (tmp10 + t3 + r) >> shift; o[7] = (tmp10 - t3 + r) >> shift;
    o[1] = (tmp11 + t2 + r) >> shift; o[6] = (tmp11 - t2 + r) >> shift;
    o[2] = (tmp12 + t1 + r) >> shift; o[5] = (tmp12 - t1 + r) >> shift;
    o[3] = (tmp13 + t0 + r) >> shift; o[4] = (tmp13 - t0 + r) >> shift;
}

__device__ __forceinline__ uint32_t jpeg_range_limit(int32_t v)       // IJG range_limit[v & RANGE_MASK], +128 level shift
{
    const uint32_t x = (uint32_t)v & 1023u;
    return x < 128u ? x + 128u : (x < 512u ? 255u : (x < 896u ? 0u : x - 896u));
}

// grid (ceil(max_blocks / 32), n); 256 threads = 32 blocks x 8 lanes
__global__ __launch_bounds__(256) void k_jpeg_idct(JpegArgs a)
{
    __shared__ int32_t s_ws[32][8][9];
    const cama_jpeg_image &D = a.imgs[blockIdx.y];
    const uint32_t lane = threadIdx.x & 7u, lb = threadIdx.x >> 3;
    const uint32_t j = blockIdx.x * 32u + lb;                    // scan-order block
    const bool live = j < D.total_blocks && D.kind != CAMA_JPEG_SEGMENT;
    uint32_t ci = 0, bx = 0, by = 0;
    if (live) {
        const uint32_t mcu = j / D.bpm, within = j - mcu * D.bpm, luma = D.hs * D.vs;
        const uint32_t mcx = mcu % D.mx, mcy = mcu / D.mx;
        if (within < luma) {
            bx = mcx * D.hs + within % D.hs;
            by = mcy * D.vs + within / D.hs;
        } else {
            ci = within - luma + 1u;
            bx = mcx;
            by = mcy;
        }
        // pass 1: column `lane`
        const int16_t *cf = a.coef + D.coef_off + (size_t)j * 64;
        const uint16_t *q = a.quant + ((size_t)D.quant_set * 3 + ci) * 64;
        int32_t d[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) d[r] = (int32_t)cf[r * 8 + lane] * (int32_t)q[r * 8 + lane];
        if (lane == 0u) d[0] = (int32_t)a.dcd[D.coef_off / 64 + j] * (int32_t)q[0];      // the DC term lives in the compact array
        jpeg_idct8(d, o, 13 - 2);
#pragma unroll
        for (int r = 0; r < 8; ++r) s_ws[lb][r][lane] = o[r];
    }
    __syncthreads();
    if (!live) return;
    int32_t d[8], o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = s_ws[lb][lane][k];
    jpeg_idct8(d, o, 13 + 2 + 3);
    uint32_t p0 = 0, p1 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p0 |= jpeg_range_limit(o[k]) << (8 * k);
        p1 |= jpeg_range_limit(o[k + 4]) << (8 * k);
    }
    uint8_t *plane = a.planes + D.plane_off[ci];
    *reinterpret_cast<uint2 *>(plane + (size_t)(by * 8u + lane) * D.plane_w[ci] + bx * 8u) = make_uint2(p0, p1);
}

// ------------------------------------------------------------------------------------------ upsampling + colour
// 8 consecutive chroma-upsampled samples of one output row starting at x0 (multiple of 8), from dword loads:
// the triangle filters need chroma columns x0/2-1 .. x0/2+4 of one (h2v1) or two (h2v2) rows.
__device__ __forceinline__ void jpeg_chroma8(const uint8_t *p, uint32_t pw, uint32_t cw, uint32_t ch, uint32_t hs,
                                             uint32_t vs, uint32_t x0, uint32_t y, int32_t out[8])
{
    if (hs == 1u) {
        const uint2 v = *reinterpret_cast<const uint2 *>(p + (size_t)y * pw + x0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            out[k] = (int32_t)((v.x >> (8 * k)) & 255u);
            out[k + 4] = (int32_t)((v.y >> (8 * k)) & 255u);
        }
        return;
    }
    const uint32_t cx0 = x0 >> 1;                                   // multiple of 4
    const uint32_t cy = vs == 2u ? y >> 1 : y;
    if (cw <= 2u) {
        // libjpeg-turbo uses the fancy routines only for downsampled widths > 2 (jdsample.c); narrower components
        // are replicated
        const uint32_t v = *reinterpret_cast<const uint32_t *>(p + (size_t)cy * pw + cx0);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[2 * k] = out[2 * k + 1] = (int32_t)((v >> (8 * k)) & 255u);
        return;
    }
    const uint32_t ny = vs == 2u ? ((y & 1u) ? min(cy + 1u, ch - 1u) : (cy ? cy - 1u : 0u)) : cy;
    const uint8_t *r0 = p + (size_t)cy * pw, *r1 = p + (size_t)ny * pw;
    const uint32_t lo = cx0 ? cx0 - 4u : 0u, hi = cx0 + 4u < pw ? cx0 + 4u : cx0;
    const uint32_t a0 = *reinterpret_cast<const uint32_t *>(r0 + lo), a1 = *reinterpret_cast<const uint32_t *>(r0 + cx0),
                   a2 = *reinterpret_cast<const uint32_t *>(r0 + hi);
    const uint32_t b0 = *reinterpret_cast<const uint32_t *>(r1 + lo), b1 = *reinterpret_cast<const uint32_t *>(r1 + cx0),
                   b2 = *reinterpret_cast<const uint32_t *>(r1 + hi);
    // column values s[-1..4]: h2v2 = 3 * nearer row + further row, h2v1 = the sample itself
    int32_t s[6];
    const int32_t wn = vs == 2u ? 3 : 1, wf = vs == 2u ? 1 : 0;
    s[0] = wn * (int32_t)(a0 >> 24) + wf * (int32_t)(b0 >> 24);
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k + 1] = wn * (int32_t)((a1 >> (8 * k)) & 255u) + wf * (int32_t)((b1 >> (8 * k)) & 255u);
    s[5] = wn * (int32_t)(a2 & 255u) + wf * (int32_t)(b2 & 255u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t cx = cx0 + k;
        const int32_t c = s[k + 1];
        if (vs == 2u) {
            out[2 * k] = cx == 0u ? (c * 4 + 8) >> 4 : (c * 3 + s[k] + 8) >> 4;
            out[2 * k + 1] = cx + 1u >= cw ? (c * 4 + 7) >> 4 : (c * 3 + s[k + 2] + 7) >> 4;
        } else {
            out[2 * k] = cx == 0u ? c : (c * 3 + s[k] + 1) >> 2;
            out[2 * k + 1] = cx + 1u >= cw ? c : (c * 3 + s[k + 2] + 2) >> 2;
        }
    }
}

// grid (ceil(W/2048), ceil(H/4), n), 256 threads: 8 consecutive pixels (24 bytes = three 8-byte stores when the row
// pitch allows it) of 4 rows each; planes are read with dword / 8-byte loads
constexpr int JPEG_COLOUR_ROWS = 4;
__global__ __launch_bounds__(256) void k_jpeg_colour(JpegArgs a)
{
    const cama_jpeg_image &D = a.imgs[blockIdx.z];
    const uint32_t x0 = (blockIdx.x * 256u + threadIdx.x) * 8u;
    if (x0 >= D.width || D.kind == CAMA_JPEG_SEGMENT) return;
    const uint32_t cw = (D.width + D.hs - 1u) / D.hs, ch = (D.height + D.vs - 1u) / D.vs;
    uint8_t *img = a.out + (size_t)D.out_slot * a.out_stride;
    const bool packed = (D.width & 7u) == 0u && ((uintptr_t)img & 7u) == 0u;
    for (uint32_t y = blockIdx.y * JPEG_COLOUR_ROWS; y < min((blockIdx.y + 1u) * JPEG_COLOUR_ROWS, D.height); ++y) {
        const uint2 yv = *reinterpret_cast<const uint2 *>(a.planes + D.plane_off[0] + (size_t)y * D.plane_w[0] + x0);
        int32_t cb[8], cr[8];
        if (D.ncomp == 3u) {
            jpeg_chroma8(a.planes + D.plane_off[1], D.plane_w[1], cw, ch, D.hs, D.vs, x0, y, cb);
            jpeg_chroma8(a.planes + D.plane_off[2], D.plane_w[2], cw, ch, D.hs, D.vs, x0, y, cr);
        }
        uint32_t px[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int32_t Y = (int32_t)(((k < 4 ? yv.x : yv.y) >> (8 * (k & 3))) & 255u);
            int32_t r = Y, g = Y, b = Y;
            if (D.ncomp == 3u) {
                const int32_t u = cb[k] - 128, v = cr[k] - 128;
                // IJG jdcolor: 16-bit fixed point, arithmetic shifts
                r = Y + ((91881 * v + 32768) >> 16);
                g = Y + ((-22554 * u + 32768 - 46802 * v) >> 16);
                b = Y + ((116130 * u + 32768) >> 16);
                r = min(max(r, 0), 255); g = min(max(g, 0), 255); b = min(max(b, 0), 255);
            }
            px[k] = a.bgr ? ((uint32_t)b | ((uint32_t)g << 8) | ((uint32_t)r << 16))
                          : ((uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16));
        }
        uint8_t *o = img + ((size_t)y * D.width + x0) * 3;
        if (packed) {
            uint2 *o8 = reinterpret_cast<uint2 *>(o);
            o8[0] = make_uint2(px[0] | (px[1] << 24), (px[1] >> 8) | (px[2] << 16));
            o8[1] = make_uint2((px[2] >> 16) | (px[3] << 8), px[4] | (px[5] << 24));
            o8[2] = make_uint2((px[5] >> 8) | (px[6] << 16), (px[6] >> 16) | (px[7] << 8));
        } else {
            for (uint32_t k = 0; k < 8u && x0 + k < D.width; ++k) {
                o[3 * k] = (uint8_t)px[k];
                o[3 * k + 1] = (uint8_t)(px[k] >> 8);
                o[3 * k + 2] = (uint8_t)(px[k] >> 16);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ restart markers
// Positions of every 0xFF 0xD0..0xD7 pair in bytes [0, n) (T.81 B.2.1: inside entropy-coded data 0xFF is always
// followed by 0x00, so such a pair IS a restart marker; hits in file headers between the scans are filtered by the
// caller, who knows the scan ranges).  16 bytes per thread, one counter reservation per wave; the list is unordered
// across waves (the caller sorts it: it is ~1 entry per 5 KB).  *count is the number found, even beyond `capacity`.
__global__ __launch_bounds__(256) void k_jpeg_find_restarts(const uint8_t *bytes, uint64_t n, uint32_t *positions,
                                                            uint32_t capacity, uint32_t *count)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    uint32_t hits = 0;                                  // bit k: marker starts at byte i0 + k
    if (i0 < n) {
        uint32_t w[5];
        if (i0 + 16 <= n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(bytes + i0);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
            for (int q = 0; q < 4; ++q) {
                w[q] = 0;
                for (int b = 0; b < 4; ++b)
                    if (i0 + q * 4 + b < n) w[q] |= (uint32_t)bytes[i0 + q * 4 + b] << (8 * b);
            }
        }
        w[4] = i0 + 16 < n ? bytes[i0 + 16] : 0u;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t b0 = (w[k >> 2] >> (8 * (k & 3))) & 255u;
            const uint32_t b1 = (w[(k + 1) >> 2] >> (8 * ((k + 1) & 3))) & 255u;
            if (b0 == 0xFFu && (b1 & 0xF8u) == 0xD0u) hits |= 1u << k;
        }
    }
    const uint32_t mine = __popc(hits);
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off, 64);
        if ((int)(threadIdx.x & 63) >= off) incl += v;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    if (total == 0u) return;                            // wave-uniform: almost every wave
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 63) base = atomicAdd(count, total);
    base = __shfl(base, 63, 64);
    uint32_t at = base + incl - mine;
    while (hits) {
        const int k = __ffs(hits) - 1;
        hits &= hits - 1u;
        if (at < capacity) positions[at] = (uint32_t)(i0 + k);
        ++at;
    }
}
