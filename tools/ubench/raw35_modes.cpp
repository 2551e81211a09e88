// Micro-benchmark (not part of the product): the 3:5 raw-frame overlay (the reference's default 1600x900 -> 960x540
// pipeline, cama/reproject.py:232-240) through the C ABI from a bare HIP program: classic one-band-per-workgroup kernel
// against the wave-specialised persistent one (option raw35_ws = workgroups per CU), same buffers, alternating, with a byte
// comparison of what they wrote.  Stamp-free (N = 0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/ubench/raw35_modes.cpp -o tools/ubench/raw35_modes \
//         -L cama_amd -lcama_hip -Wl,-rpath,'$ORIGIN/../../cama_amd'
//   tools/ubench/raw35_modes [frames=40] [script: comma-separated "ws:order"]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "cama_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define CA(x) do { int r_ = (x); if (r_ < 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, cama_last_error()); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int F = argc > 1 ? atoi(argv[1]) : 40;
    const std::string script = argc > 2 ? argv[2] : "0:-1,2:-1,0:-1,2:-1";
    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 20;
    const int C = 6, H0 = 900, W0 = 1600, H = 540, W = 960, cols = 3, radius = 2;
    const size_t src_bytes = (size_t)F * C * H0 * W0 * 3, dst_bytes = (size_t)F * C * H * W * 3;
    uint8_t *src, *dst, *ref;
    CK(hipMalloc(&src, src_bytes)); CK(hipMalloc(&dst, dst_bytes)); CK(hipMalloc(&ref, dst_bytes));
    {   // deterministic, non-constant source bytes
        std::vector<uint8_t> h((size_t)C * H0 * W0 * 3);
        uint32_t x = 12345;
        for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (uint8_t)(x >> 24); }
        for (int f = 0; f < F; ++f) CK(hipMemcpy(src + (size_t)f * h.size(), h.data(), h.size(), hipMemcpyHostToDevice));
    }
    // the 3:5 maps of a zero-distortion 0.6 scale: src = dst / 0.6 (cama/reproject.py:176-182, 238-239)
    std::vector<float> mx((size_t)C * W), my((size_t)C * H);
    for (int c = 0; c < C; ++c) {
        for (int x = 0; x < W; ++x) mx[(size_t)c * W + x] = (float)(x / 0.6);
        for (int y = 0; y < H; ++y) my[(size_t)c * H + y] = (float)(y / 0.6);
    }
    const int R = cama_overlay_band_rows(W), NB = (H + R - 1) / R;
    std::vector<uint32_t> vrows((size_t)C * H * 2);
    std::vector<int32_t> brows((size_t)C * NB * 2);
    int32_t most = 0;
    const int rc = cama_raw35_plan(mx.data(), my.data(), C, H, W, H0, W0, vrows.data(), brows.data(), &most);
    if (rc != 1) { fprintf(stderr, "cama_raw35_plan -> %d (%s)\n", rc, cama_last_error()); return 1; }
    uint32_t *d_vrows; int32_t *d_brows;
    CK(hipMalloc(&d_vrows, vrows.size() * 4)); CK(hipMalloc(&d_brows, brows.size() * 4));
    CK(hipMemcpy(d_vrows, vrows.data(), vrows.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_brows, brows.data(), brows.size() * 4, hipMemcpyHostToDevice));
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
    printf("# raw35 F=%d max_src_rows=%d (half bands: %d) band rows=%d reps=%d\n", F, most & 0xffff, most >> 16, R, reps);
    // reference bytes: the classic kernel
    CA(cama_set_option("raw35_ws", 0));
    CA(cama_set_option("raw35_subrows", 0));
    CA(cama_overlay_frames_raw35(src, H0, W0, d_vrows, d_brows, most, ref, 0, F, C, H, W, cols, radius, hw, pal, scratch, sb, s));
    CK(hipStreamSynchronize(s));
    std::vector<uint8_t> href(std::min(dst_bytes, (size_t)256 << 20)), hgot(href.size());
    CK(hipMemcpy(href.data(), ref, href.size(), hipMemcpyDeviceToHost));
    size_t pos = 0;
    while (pos < script.size()) {
        size_t end = script.find(',', pos);
        if (end == std::string::npos) end = script.size();
        long ws = 0, order = -1, nl = 2, stg = 0;
        sscanf(script.substr(pos, end - pos).c_str(), "%ld:%ld:%ld:%ld", &ws, &order, &nl, &stg);
        CA(cama_set_option("raw35_subrows", stg));           // 4th field: 1 = half bands, 0 = whole bands
        CA(cama_set_option("raw35_loaders", nl));
        pos = end + 1;
        CA(cama_set_option("raw35_ws", ws));
        CA(cama_set_option("overlay_chunk_log2", order));
        CK(hipMemsetAsync(dst, 0xA5, dst_bytes, s));
        const auto launch = [&] { CA(cama_overlay_frames_raw35(src, H0, W0, d_vrows, d_brows, most, dst, 0, F, C, H, W, cols, radius, hw, pal, scratch, sb, s)); };
        for (int k = 0; k < 3; ++k) launch();
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(hgot.data(), dst, hgot.size(), hipMemcpyDeviceToHost));
        const bool same = !memcmp(hgot.data(), href.data(), href.size());
        CA(cama_profile_enable(1));
        for (int k = 0; k < reps; ++k) launch();
        CK(hipStreamSynchronize(s));
        std::vector<double> ms(reps);
        int32_t got = 0;
        CA(cama_profile_collect_each(ms.data(), reps, &got));
        CA(cama_profile_enable(0));
        std::sort(ms.begin(), ms.end());
        const double bytes = (double)src_bytes + (double)dst_bytes;
        printf("ws %ld loaders %ld halves %ld order %3ld  min %.4f med %.4f max %.4f ms   frac(med) %.3f   bytes %s\n", ws, nl, stg, order, ms[0], ms[reps / 2], ms[reps - 1],
               bytes / (ms[reps / 2] * 1e-3) / 8e12, same ? "identical to the classic kernel" : "DIFFER");
        fflush(stdout);
    }
    return 0;
}
