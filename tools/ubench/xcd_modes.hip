// Micro-benchmark (not part of the product): the per-process speed modes of an XCD-contiguous streaming copy, outside
// torch and outside libcama_hip.  2 x 1.04 GB (one headline step's frames and mosaic), one 16-byte chunk per thread,
// non-temporal; workgroup L copies chunk-block L ("interleaved": all XCDs advance through one stream) or block
// (L % 8) * per + L / 8 ("contiguous": XCD x streams its own eighth).  Run it several times in a row:
//   hipcc --offload-arch=gfx950 -O3 xcd_modes.hip -o xcd_modes && for i in 1 2 3 4 5 6 7 8; do ./xcd_modes; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool CONTIG>
__global__ __launch_bounds__(256) void k_copy(const u32x4 *__restrict__ s, u32x4 *__restrict__ d, size_t n, unsigned per)
{
    const unsigned L = blockIdx.x;
    const size_t blk = CONTIG ? (size_t)(L & 7u) * per + (L >> 3) : (size_t)L;
    const size_t i = blk * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}

static hipStream_t g_stream = 0;
template <typename F>
static double timeit(F f)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int k = 0; k < 5; ++k) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, g_stream));
    for (int k = 0; k < 20; ++k) f();
    CK(hipEventRecord(b, g_stream));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 20.0;
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)40 * 6 * 900 * 1600 * 3, n = bytes / 16;
    const int pairs = argc > 1 ? atoi(argv[1]) : 1;
    const unsigned blocks = (unsigned)((n + 255) / 256), per = (blocks + 7) / 8, grid8 = per * 8;
    for (int p = 0; p < pairs; ++p) {                    // (earlier pairs stay allocated: every pair sits on other pages)
        u32x4 *s, *d; CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes));
        CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 2, bytes));
        const double ti = timeit([&] { hipLaunchKernelGGL((k_copy<false>), dim3(blocks), dim3(256), 0, g_stream, s, d, n, per); });
        const double tc = timeit([&] { hipLaunchKernelGGL((k_copy<true>), dim3(grid8), dim3(256), 0, g_stream, s, d, n, per); });
        const double ti2 = timeit([&] { hipLaunchKernelGGL((k_copy<false>), dim3(blocks), dim3(256), 0, g_stream, s, d, n, per); });
        const double tc2 = timeit([&] { hipLaunchKernelGGL((k_copy<true>), dim3(grid8), dim3(256), 0, g_stream, s, d, n, per); });
        printf("%ssrc %p dst %p   interleaved %.3f / %.3f   contiguous %.3f / %.3f   of 8 TB/s\n", p ? "   +  " : "", (void *)s, (void *)d,
               2.0 * bytes / ti / 8e9, 2.0 * bytes / ti2 / 8e9, 2.0 * bytes / tc / 8e9, 2.0 * bytes / tc2 / 8e9);
        if (argc > 2) {                                   // the same pair on a few more streams (other hardware queues)
            printf("      contiguous on %d more streams:", atoi(argv[2]));
            for (int q = 0; q < atoi(argv[2]); ++q) {
                CK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
                const double t = timeit([&] { hipLaunchKernelGGL((k_copy<true>), dim3(grid8), dim3(256), 0, g_stream, s, d, n, per); });
                printf(" %.3f", 2.0 * bytes / t / 8e9);
            }
            printf("\n");
            g_stream = 0;
        }
    }
    return 0;
}
