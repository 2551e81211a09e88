// Micro-benchmark (not part of the product): the overlay kernel of libcama_hip.so driven through the C ABI from a bare HIP
// program -- no Python, no torch -- to study what decides its bandwidth: how the buffers were allocated (hipMalloc,
// physically contiguous, one big arena), the workgroup -> band order, the stagger between the eight XCD streams, the
// translation look-ahead, and whether a launch touches buffers it has touched before (warm) or new ones (cold).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/ubench/overlay_modes.cpp -o tools/ubench/overlay_modes \
//         -L cama_amd -lcama_hip -Wl,-rpath,'$ORIGIN/../../cama_amd'
//   tools/ubench/overlay_modes <alloc: malloc|contig|arena> <frames per launch> <sets> <script>
// script = comma-separated runs "order:rot:prefetch:cold" (order -1 = library's choice, cold 0/1), each timed over `reps`
// launches with per-launch events.  Output: one line per run with min / median / mean / max launch time and the fraction
// of 8 TB/s (36*W*H*F bytes per launch).  Stamp-free (N = 0): the copy structure is what is being studied.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "cama_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define CA(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, cama_last_error()); exit(1); } } while (0)

// "vmm:<chunk MB>:<shuffle>": one virtual range backed by separate physical allocations of <chunk MB> each
// (hipMemCreate), mapped in allocation order (shuffle 0), in a seeded random order (1) or reversed (2): what the physical
// placement of a buffer's pieces does to the eight XCD streams, under this program's control.
static unsigned g_seed = 12345;
static void *alloc_vmm(size_t bytes, size_t chunk, int shuffle, size_t va_align)
{
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk;
    void *base = nullptr;
    CK(hipMemAddressReserve(&base, n * chunk, va_align ? va_align : (2u << 20), nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    for (size_t k = 0; k < n; ++k) CK(hipMemCreate(&h[k], chunk, &prop, 0));
    std::vector<size_t> order(n);
    for (size_t k = 0; k < n; ++k) order[k] = shuffle == 2 ? n - 1 - k : k;
    if (shuffle == 1)
        for (size_t k = n; k > 1; --k) {
            g_seed = g_seed * 1664525u + 1013904223u;
            std::swap(order[k - 1], order[(g_seed >> 8) % k]);
        }
    for (size_t k = 0; k < n; ++k) CK(hipMemMap((char *)base + k * chunk, chunk, 0, h[order[k]], 0));
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(base, n * chunk, &acc, 1));
    return base;
}

static void *alloc_dev(const char *kind, size_t bytes)
{
    void *p = nullptr;
    if (!strncmp(kind, "vmm", 3)) {
        double mb = 2, va_mb = 0;
        int shuffle = 0;
        sscanf(kind, "vmm:%lf:%d:%lf", &mb, &shuffle, &va_mb);                 // vmm:<chunk MB>:<shuffle>[:<VA alignment MB>]
        return alloc_vmm(bytes, (size_t)(mb * (1 << 20)), shuffle, (size_t)(va_mb * (1 << 20)));
    }
    if (!strcmp(kind, "contig")) CK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous));
    else CK(hipMalloc(&p, bytes));
    return p;
}

int main(int argc, char **argv)
{
    const char *kind = argc > 1 ? argv[1] : "malloc";
    const int F = argc > 2 ? atoi(argv[2]) : 40;
    const int sets = argc > 3 ? atoi(argv[3]) : 1;             // (frames, mosaic) pairs; cold runs cycle through them
    const std::string script = argc > 4 ? argv[4] : "-1:0:0:0";
    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 24;
    const int C = 6, H = 900, W = 1600, cols = 3, radius = 2;
    const size_t frame_bytes = (size_t)C * H * W * 3, set_bytes = frame_bytes * F;
    std::vector<uint8_t *> src(sets), dst(sets);
    if (!strcmp(kind, "arena")) {                                // one allocation, carved: what a caching allocator hands out
        uint8_t *a = (uint8_t *)alloc_dev("malloc", set_bytes * 2 * sets);
        for (int k = 0; k < sets; ++k) { src[k] = a + (size_t)2 * k * set_bytes; dst[k] = src[k] + set_bytes; }
    } else {
        for (int k = 0; k < sets; ++k) { src[k] = (uint8_t *)alloc_dev(kind, set_bytes); dst[k] = (uint8_t *)alloc_dev(kind, set_bytes); }
    }
    for (int k = 0; k < sets; ++k) { CK(hipMemset(src[k], 17 + k, set_bytes)); CK(hipMemset(dst[k], 0, set_bytes)); }
    const size_t sb = cama_render_scratch_bytes(0, F, C, H, W, radius);
    void *scratch; CK(hipMalloc(&scratch, sb));
    double *w2c, *c2cam, *K;
    CK(hipMalloc(&w2c, (size_t)F * 128)); CK(hipMalloc(&c2cam, C * 128)); CK(hipMalloc(&K, C * 72));
    CK(hipMemset(w2c, 0, (size_t)F * 128)); CK(hipMemset(c2cam, 0, C * 128)); CK(hipMemset(K, 0, C * 72));
    const double crop[6] = {-50, 50, -100, 100, -200, 200};
    int32_t hw[16];
    cama_circle_halfwidths(radius, hw);
    const uint8_t pal[6] = {211, 211, 211, 0, 215, 255};
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CA(cama_bin_frames(nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, w2c, F, c2cam, K, C, crop, W, H, radius, scratch, sb, s));
    CK(hipStreamSynchronize(s));
    printf("# alloc=%s F=%d sets=%d reps=%d  src0=%p dst0=%p\n", kind, F, sets, reps, (void *)src[0], (void *)dst[0]);
    size_t pos = 0;
    int cursor = 0;                                             // cold runs keep walking the ring of sets across runs
    while (pos < script.size()) {
        size_t end = script.find(',', pos);
        if (end == std::string::npos) end = script.size();
        long order = -1, rot = 0, pf = 0, cold = 0, io = 0, grp = 0, si = -1, di = -1;
        sscanf(script.substr(pos, end - pos).c_str(), "%ld:%ld:%ld:%ld:%ld:%ld:%ld:%ld", &order, &rot, &pf, &cold, &io, &grp, &si, &di);
        CA(cama_set_option("overlay_item_order", io));
        CA(cama_set_option("overlay_groups_log2", grp));
        pos = end + 1;
        CA(cama_set_option("overlay_chunk_log2", order));
        CA(cama_set_option("overlay_tune", 0));
        CA(cama_set_option("overlay_rot", rot));
        CA(cama_set_option("overlay_prefetch", pf));
        const auto launch = [&](int k) {          // (fields 7, 8 of a run: a fixed source set and a fixed mosaic set)
            CA(cama_overlay_frames(src[si >= 0 ? si % sets : k], dst[di >= 0 ? di % sets : k], 0, F, C, H, W, cols, radius, hw, pal,
                                   scratch, sb, s));
        };
        for (int k = 0; k < 3; ++k) launch(cold ? (cursor++ % sets) : 0);       // warm-up (cold: just moves on)
        CK(hipStreamSynchronize(s));
        CA(cama_profile_enable(1));                               // the kernel's own start / stop events
        for (int k = 0; k < reps; ++k) launch(cold ? (cursor++ % sets) : 0);
        CK(hipStreamSynchronize(s));
        std::vector<double> ms(reps);
        int32_t got = 0;
        CA(cama_profile_collect_each(ms.data(), reps, &got));
        CA(cama_profile_enable(0));
        if (got != reps) { fprintf(stderr, "expected %d timed launches, got %d\n", reps, got); exit(1); }
        std::vector<double> so = ms;
        std::sort(so.begin(), so.end());
        double mean = 0;
        for (double v : ms) mean += v;
        mean /= reps;
        const double bytes = 2.0 * set_bytes;
        if (si >= 0 || di >= 0) printf("src %ld dst %ld  ", si, di);
        printf("order %3ld grp %ld io %ld rot %5ld pf %4ld %s  min %.4f med %.4f mean %.4f max %.4f ms   frac(med) %.3f frac(mean) %.3f\n", order, grp, io, rot, pf,
               cold ? "cold" : "warm", so[0], so[reps / 2], mean, so[reps - 1], bytes / (so[reps / 2] * 1e-3) / 8e12,
               bytes / (mean * 1e-3) / 8e12);
        fflush(stdout);
    }
    return 0;
}
