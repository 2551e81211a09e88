// cama_jpeg.hip -- the device baseline-JPEG decoder of libcama_hip.so (frame ingest; the reference decodes every
// frame with cv2.imread = libjpeg(-turbo) on one host core, cama/reproject.py:224,243) and the file-reader helper of the
// ingest path.  Kernels: jpeg_kernels.hpp.  Entry points: cama_jpeg_* and cama_read_files (include/cama_hip.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cama_common.hpp"

namespace {
using cama_impl::align_up;
using cama_impl::fail;

#include "jpeg_kernels.hpp"

// ------------------------------------------------------------------------------------------ device JPEG decode
struct JpegLayout {
    size_t clean, tile_count, tile_base, nbits, E, nb, wg_total, coef, dcd, planes, total;
    size_t coef_elems, plane_bytes, clean_bytes;
    uint32_t total_wgs, total_tiles, max_blocks;
};

// derived descriptor fields + scratch layout; `write` = fill the [plan] fields (cama_jpeg_plan) or check them
int jpeg_layout(cama_jpeg_image *imgs, const cama_jpeg_image *cimgs, int32_t n, uint64_t stream_bytes, bool write,
                       JpegLayout &L)
{
    if (n < 1 || n > 65535) return fail(CAMA_EINVAL, "n=%d images out of range [1, 65535]", n);
    uint32_t wg = 0, tile = 0, max_blocks = 0;
    size_t coef = 0, planes = 0, clean = 0;
    std::vector<cama_jpeg_image> planned((size_t)n);
    for (int i = 0; i < n; ++i) {
        const cama_jpeg_image &D = cimgs[i];
        if (D.width < 1 || D.width > 65535 || D.height < 1 || D.height > 65535)
            return fail(CAMA_EINVAL, "image %d: %ux%u out of range", i, D.width, D.height);
        if (!(D.ncomp == 1 || D.ncomp == 3)) return fail(CAMA_EINVAL, "image %d: %u components", i, D.ncomp);
        const bool samp_ok = (D.hs == 1 && D.vs == 1) || (D.ncomp == 3 && D.hs == 2 && (D.vs == 1 || D.vs == 2));
        if (!samp_ok) return fail(CAMA_EINVAL, "image %d: sampling %ux%u not supported", i, D.hs, D.vs);
        if (D.kind > CAMA_JPEG_PIXELS) return fail(CAMA_EINVAL, "image %d: kind %u", i, D.kind);
        for (uint32_t c = 0; c < D.ncomp; ++c)
            if (D.comp_dc[c] > 1 || D.comp_ac[c] > 1) return fail(CAMA_EINVAL, "image %d: Huffman selector > 1", i);
        const bool has_stream = D.kind != CAMA_JPEG_PIXELS, has_pixels = D.kind != CAMA_JPEG_SEGMENT;
        if (has_stream) {
            if (D.stream_len < 1 || D.stream_len > (1u << 29) || D.stream_off + D.stream_len > stream_bytes)
                return fail(CAMA_EINVAL, "image %d: segment [%llu, +%u) outside the %llu stream bytes", i,
                            (unsigned long long)D.stream_off, D.stream_len, (unsigned long long)stream_bytes);
        } else if (D.stream_len != 0) {
            return fail(CAMA_EINVAL, "image %d: a pixels-only descriptor carries no stream", i);
        }
        cama_jpeg_image W = D;
        W.clean_off = has_stream ? clean : 0;
        if (has_stream) clean += align_up((size_t)D.stream_len + 64, 16);   // zero slack after every unstuffed segment
        W.mx = (D.width + 8 * D.hs - 1) / (8 * D.hs);
        W.my = (D.height + 8 * D.vs - 1) / (8 * D.vs);
        W.bpm = D.ncomp == 1 ? 1u : D.hs * D.vs + 2u;
        const uint64_t blocks = (uint64_t)W.mx * W.my * W.bpm;
        if (blocks > (1u << 28)) return fail(CAMA_EINVAL, "image %d: too many blocks", i);
        W.total_blocks = (uint32_t)blocks;
        const uint32_t nsub = (uint32_t)(((uint64_t)D.stream_len * 8 + JPEG_SUB_BITS - 1) / JPEG_SUB_BITS);
        W.wg0 = wg;
        W.nwg = (nsub + JPEG_WG - 1) / JPEG_WG;
        W.tile0 = tile;
        W.ntile = (D.stream_len + JPEG_TILE - 1) / JPEG_TILE;
        if (D.kind == CAMA_JPEG_SEGMENT) {
            if (D.parent >= (uint32_t)i) return fail(CAMA_EINVAL, "image %d: parent %u must come earlier", i, D.parent);
            const cama_jpeg_image &P = planned[D.parent];
            if (P.kind != CAMA_JPEG_PIXELS || P.ncomp != D.ncomp || P.hs != D.hs || P.vs != D.vs || W.my != 1 ||
                (uint64_t)D.first_block + blocks > P.total_blocks || D.first_block % W.bpm)
                return fail(CAMA_EINVAL, "image %d: restart segment does not fit its parent %u", i, D.parent);
            W.coef_off = P.coef_off + (uint64_t)D.first_block * 64;
            W.out_slot = 0;
        } else {
            W.coef_off = coef;
            coef += (size_t)blocks * 64;
            W.parent = 0; W.first_block = 0;
            if (D.out_slot >= (uint32_t)n) return fail(CAMA_EINVAL, "image %d: out_slot %u out of range", i, D.out_slot);
        }
        for (uint32_t c = 0; c < 3; ++c) {
            W.plane_off[c] = 0; W.plane_w[c] = 0; W.plane_h[c] = 0;
            if (c < D.ncomp && has_pixels) {
                W.plane_w[c] = W.mx * (c == 0 ? D.hs : 1u) * 8u;
                W.plane_h[c] = W.my * (c == 0 ? D.vs : 1u) * 8u;
                W.plane_off[c] = planes;
                planes += align_up((size_t)W.plane_w[c] * W.plane_h[c], 16);
            }
        }
        wg += W.nwg;
        tile += W.ntile;
        if (has_pixels) max_blocks = std::max(max_blocks, W.total_blocks);
        planned[i] = W;
        if (write) imgs[i] = W;
        else if (memcmp(&W, &D, sizeof(W)) != 0)
            return fail(CAMA_EINVAL, "image %d: descriptor was not produced by cama_jpeg_plan()", i);
    }
    L.total_wgs = wg; L.total_tiles = tile; L.max_blocks = max_blocks;
    L.coef_elems = coef; L.plane_bytes = planes; L.clean_bytes = clean + 64;
    size_t off = 0;
    L.clean = off;      off = align_up(off + clean + 64, 256);
    L.tile_count = off; off = align_up(off + (size_t)tile * 4, 256);
    L.tile_base = off;  off = align_up(off + (size_t)tile * 4, 256);
    L.nbits = off;      off = align_up(off + (size_t)n * 4, 256);
    L.E = off;          off = align_up(off + (size_t)wg * JPEG_WG * 8, 256);
    L.nb = off;         off = align_up(off + (size_t)wg * JPEG_WG * 4, 256);
    L.wg_total = off;   off = align_up(off + (size_t)wg * 4, 256);
    L.coef = off;       off = align_up(off + coef * 2, 256);
    L.dcd = off;        off = align_up(off + coef / 64 * 2 + 64, 256);
    L.planes = off;     off = align_up(off + planes, 256);
    L.total = off;
    return CAMA_OK;
}

}  // namespace

extern "C" size_t cama_jpeg_image_bytes(void) { return sizeof(cama_jpeg_image); }
extern "C" size_t cama_jpeg_huff_set_bytes(void) { return sizeof(JpegHuffRec); }

extern "C" int cama_jpeg_plan(cama_jpeg_image *imgs, int32_t n, uint64_t stream_bytes, cama_jpeg_plan_info *info)
{
    if (!imgs || !info) return fail(CAMA_EINVAL, "NULL pointer argument");
    JpegLayout L;
    if (int rc = jpeg_layout(imgs, imgs, n, stream_bytes, true, L)) return rc;
    info->scratch_bytes = L.total;
    info->total_wgs = L.total_wgs;
    info->total_tiles = L.total_tiles;
    info->max_blocks = L.max_blocks;
    info->reserved = 0;
    return CAMA_OK;
}

extern "C" int cama_jpeg_find_restarts(const uint8_t *stream, uint64_t stream_bytes, uint32_t *positions,
                                       uint32_t capacity, uint32_t *count, void *stream_handle)
{
    if (!stream || !positions || !count) return fail(CAMA_EINVAL, "NULL pointer argument");
    if (stream_bytes < 1 || stream_bytes > 0xffffffffull)
        return fail(CAMA_EINVAL, "stream_bytes=%llu out of range [1, 2^32)", (unsigned long long)stream_bytes);
    if ((uintptr_t)stream % 16) return fail(CAMA_EINVAL, "stream must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream_handle;
    HIP_TRY(hipMemsetAsync(count, 0, 4, s));
    const uint64_t threads = (stream_bytes + 15) / 16;
    hipLaunchKernelGGL(k_jpeg_find_restarts, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, stream, stream_bytes,
                       positions, capacity, count);
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}

extern "C" int cama_jpeg_decode(const uint8_t *stream, uint64_t stream_bytes, const cama_jpeg_image *imgs,
                                const cama_jpeg_image *imgs_dev, int32_t n, const void *huff_sets, int32_t n_huff_sets,
                                const uint16_t *quant_sets, int32_t n_quant_sets, uint8_t *out, uint64_t out_stride,
                                int32_t bgr, void *scratch, size_t scratch_bytes, int32_t *status, void *stream_handle)
{
    if (!stream || !imgs || !imgs_dev || !huff_sets || !quant_sets || !out || !scratch || !status)
        return fail(CAMA_EINVAL, "NULL pointer argument");
    JpegLayout L;
    if (int rc = jpeg_layout(nullptr, imgs, n, stream_bytes, false, L)) return rc;
    if (scratch_bytes < L.total) return fail(CAMA_EINVAL, "scratch too small: %zu < %zu", scratch_bytes, L.total);
    uint32_t maxw = 0, maxh = 0;
    // the pixel stages (IDCT, colour) run over the descriptors that own pixels; restart-interval segments do not, and
    // there are ~60 of them per image: when they all come after the pixel owners (as cama_amd/jpeg.py lays them out)
    // the grids stop at the first one instead of launching a million workgroups that return at once
    int npix = 0;
    while (npix < n && imgs[npix].kind != CAMA_JPEG_SEGMENT) ++npix;
    for (int i = npix; i < n; ++i)
        if (imgs[i].kind != CAMA_JPEG_SEGMENT) { npix = n; break; }
    for (int i = 0; i < n; ++i) {
        if ((int32_t)imgs[i].huff_set >= n_huff_sets || (int32_t)imgs[i].quant_set >= n_quant_sets)
            return fail(CAMA_EINVAL, "image %d: table set index out of range", i);
        if (imgs[i].kind != CAMA_JPEG_SEGMENT && (uint64_t)imgs[i].width * imgs[i].height * 3 > out_stride)
            return fail(CAMA_EINVAL, "image %d: %ux%ux3 bytes exceed out_stride %llu", i, imgs[i].width, imgs[i].height,
                        (unsigned long long)out_stride);
        if (imgs[i].kind != CAMA_JPEG_SEGMENT) {
            maxw = std::max(maxw, imgs[i].width);
            maxh = std::max(maxh, imgs[i].height);
        }
    }
    if ((uintptr_t)scratch % 256 || (uintptr_t)huff_sets % 16)
        return fail(CAMA_EINVAL, "huff_sets must be 16-byte and scratch 256-byte aligned");
    hipStream_t s = (hipStream_t)stream_handle;
    char *base = (char *)scratch;
    JpegArgs a{};
    a.stream = stream; a.clean = (uint8_t *)(base + L.clean); a.imgs = imgs_dev; a.n = n;
    a.huff = (const JpegHuffRec *)huff_sets; a.quant = quant_sets;
    a.tile_count = (uint32_t *)(base + L.tile_count); a.tile_base = (uint32_t *)(base + L.tile_base);
    a.nbits = (uint32_t *)(base + L.nbits); a.E = (uint64_t *)(base + L.E); a.nb = (uint32_t *)(base + L.nb);
    a.wg_total = (uint32_t *)(base + L.wg_total); a.coef = (int16_t *)(base + L.coef); a.dcd = (int16_t *)(base + L.dcd);
    a.planes = (uint8_t *)(base + L.planes); a.out = out; a.out_stride = (size_t)out_stride; a.bgr = bgr;
    a.status = status;
    // one fill in front of the chain: the coefficients (only non-zero ones are stored).  The unstuffed copy's slack is cleared
    // by k_jpeg_unstuff, the status words by k_jpeg_tilescan, and every block's DC difference is stored by k_jpeg_write.
    HIP_TRY(hipMemsetAsync(a.coef, 0, L.coef_elems * 2, s));
    if (L.total_tiles) hipLaunchKernelGGL(k_jpeg_count, dim3(L.total_tiles), dim3(JPEG_TILE / 4), 0, s, a);
    hipLaunchKernelGGL(k_jpeg_tilescan, dim3((unsigned)n), dim3(256), 0, s, a);
    if (L.total_tiles) hipLaunchKernelGGL(k_jpeg_unstuff, dim3(L.total_tiles), dim3(JPEG_TILE / 4), 0, s, a);
    if (L.total_wgs) {
        hipLaunchKernelGGL(k_jpeg_sync<1>, dim3(L.total_wgs), dim3(JPEG_WG), 0, s, a);
        hipLaunchKernelGGL(k_jpeg_sync<2>, dim3(L.total_wgs), dim3(JPEG_WG), 0, s, a);
        hipLaunchKernelGGL(k_jpeg_write, dim3(L.total_wgs), dim3(JPEG_WG), 0, s, a);
    }
    hipLaunchKernelGGL(k_jpeg_dc, dim3((unsigned)n, 3), dim3(JPEG_DC_THREADS), 0, s, a);
    if (npix) {
        hipLaunchKernelGGL(k_jpeg_idct, dim3((L.max_blocks + 31) / 32, (unsigned)npix), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_jpeg_colour,
                           dim3((maxw + 2047) / 2048, (maxh + JPEG_COLOUR_ROWS - 1) / JPEG_COLOUR_ROWS, (unsigned)npix),
                           dim3(256), 0, s, a);
    }
    HIP_TRY(hipGetLastError());
    return CAMA_OK;
}


// ------------------------------------------------------------------------------------------
// host-side ingest helper: many files -> caller-provided (pinned) buffers, without the interpreter
// ------------------------------------------------------------------------------------------

#include <atomic>
#include <thread>
#include <fcntl.h>
#include <unistd.h>

extern "C" {

int cama_read_files(const char *const *paths, void *const *dst, const uint64_t *sizes, int32_t n, int32_t threads,
                    int32_t *status)
{
    if (n < 0 || (n && (!paths || !dst || !sizes || !status))) return fail(CAMA_EINVAL, "NULL pointer argument");
    if (n == 0) return CAMA_OK;
    const int workers = std::max(1, std::min<int>(threads, std::min(n, 64)));
    std::atomic<int> next{0};
    const auto work = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            status[i] = 1;
            if (!paths[i] || !dst[i]) continue;
            const int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
            if (fd < 0) continue;
            uint64_t got = 0;
            bool ok = true;
            while (got < sizes[i]) {
                const ssize_t k = read(fd, (char *)dst[i] + got, (size_t)(sizes[i] - got));
                if (k <= 0) { ok = false; break; }
                got += (uint64_t)k;
            }
            if (ok) {                                   // the size came from a directory scan: the file must end here
                char extra;
                ok = read(fd, &extra, 1) == 0;
            }
            close(fd);
            status[i] = ok ? 0 : 1;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < workers; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    return CAMA_OK;
}

}  // extern "C"
