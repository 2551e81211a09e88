// Micro-benchmark (not part of the product): what does a plain linear HBM->HBM copy reach on this chip with the
// same 16-byte non-temporal accesses the overlay kernel uses?  hipcc --offload-arch=gfx950 -O3 copy_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n)
{
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) { size_t i = base + (size_t)j * 256; if (i < n) v[j] = NT ? __builtin_nontemporal_load(s + i) : s[i]; }
#pragma unroll
    for (int j = 0; j < U; ++j) { size_t i = base + (size_t)j * 256; if (i < n) { if (NT) __builtin_nontemporal_store(v[j], d + i); else d[i] = v[j]; } }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_persist(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n)
{
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += (size_t)gridDim.x * 256 * U) {
        u32x4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) { size_t i = base + (size_t)j * 256; if (i < n) v[j] = NT ? __builtin_nontemporal_load(s + i) : s[i]; }
#pragma unroll
        for (int j = 0; j < U; ++j) { size_t i = base + (size_t)j * 256; if (i < n) { if (NT) __builtin_nontemporal_store(v[j], d + i); else d[i] = v[j]; } }
    }
}
// one workgroup per image row, one 16-byte chunk per thread, destination in 2x3 mosaic layout
template <int LDS_BYTES>
__global__ __launch_bounds__(320) void k_copy_mosaic_rows(const u32x4* __restrict__ s, u32x4* __restrict__ d, int C, int H, int cpr)
{
    __shared__ unsigned pad[LDS_BYTES / 4 + 1];
    if (LDS_BYTES > 4 && threadIdx.x == 1023) pad[threadIdx.x] = 1;    // keep the allocation
    const unsigned row = blockIdx.x;                  // (f*C + c)*H + y
    const unsigned fc = row / H, y = row - fc * H;
    const unsigned f = fc / C, c = fc - f * C;
    const unsigned col = threadIdx.x;
    if (col >= (unsigned)cpr) return;
    u32x4 v = __builtin_nontemporal_load(s + (size_t)row * cpr + col);
    const size_t drow = ((size_t)f * 2 * H + (size_t)(c / 3) * H + y) * 3 * cpr + (size_t)(c % 3) * cpr;
    __builtin_nontemporal_store(v, d + drow + col);
}
// band workgroups (R rows, 256 threads, U chunks in flight) with mosaic destination, no stamp logic
template <int U, int LDS_BYTES>
__global__ __launch_bounds__(256) void k_copy_mosaic_bands(const u32x4* __restrict__ s, u32x4* __restrict__ d, int C, int H, int cpr, int R, int NB)
{
    __shared__ unsigned pad[LDS_BYTES / 4 + 1];
    if (LDS_BYTES > 4 && threadIdx.x == 1023) pad[threadIdx.x] = 1;
    const unsigned bin = blockIdx.x;
    const unsigned fc = bin / NB, b = bin - fc * NB;
    const unsigned f = fc / C, c = fc - f * C;
    const int y0 = b * R, nrows = min(R, H - y0);
    const unsigned nchunks = nrows * cpr;
    const u32x4* sb = s + ((size_t)fc * H + y0) * cpr;
    u32x4* db = d + ((size_t)f * 2 * H + (size_t)(c / 3) * H + y0) * 3 * cpr + (size_t)(c % 3) * cpr;
    for (unsigned base = threadIdx.x; base < nchunks; base += 256 * U) {
        u32x4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) { unsigned i = base + j * 256; if (i < nchunks) v[j] = __builtin_nontemporal_load(sb + i); }
#pragma unroll
        for (int j = 0; j < U; ++j) { unsigned i = base + j * 256; if (i < nchunks) { unsigned r = i / cpr, col = i - r * cpr; __builtin_nontemporal_store(v[j], db + (size_t)r * 3 * cpr + col); } }
    }
}
// as k_copy_mosaic_bands but workgroups ordered (frame, camera row, band, camera column): the 3 cameras that share
// R mosaic rows are adjacent in launch order
template <int U>
__global__ __launch_bounds__(256) void k_copy_mosaic_bands_rowmajor(const u32x4* __restrict__ s, u32x4* __restrict__ d, int C, int H, int cpr, int R, int NB)
{
    unsigned t = blockIdx.x;
    const unsigned cc = t % 3; t /= 3;
    const unsigned b = t % NB; t /= NB;
    const unsigned cr = t % 2; const unsigned f = t / 2;
    const unsigned c = cr * 3 + cc, fc = f * C + c;
    const int y0 = b * R, nrows = min(R, H - y0);
    const unsigned nchunks = nrows * cpr;
    const u32x4* sb = s + ((size_t)fc * H + y0) * cpr;
    u32x4* db = d + ((size_t)f * 2 * H + (size_t)cr * H + y0) * 3 * cpr + (size_t)cc * cpr;
    for (unsigned base = threadIdx.x; base < nchunks; base += 256 * U) {
        u32x4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) { unsigned i = base + j * 256; if (i < nchunks) v[j] = __builtin_nontemporal_load(sb + i); }
#pragma unroll
        for (int j = 0; j < U; ++j) { unsigned i = base + j * 256; if (i < nchunks) { unsigned r = i / cpr, col = i - r * cpr; __builtin_nontemporal_store(v[j], db + (size_t)r * 3 * cpr + col); } }
    }
}
// one workgroup per (frame, camera row, band): 3 x 256 threads, slice k copies camera column k's band, so the
// workgroup as a whole writes R CONTIGUOUS mosaic rows
template <int U>
__global__ __launch_bounds__(768) void k_copy_mosaic_fullrow(const u32x4* __restrict__ s, u32x4* __restrict__ d, int C, int H, int cpr, int R, int NB)
{
    unsigned t = blockIdx.x;
    const unsigned b = t % NB; t /= NB;
    const unsigned cr = t % 2; const unsigned f = t / 2;
    const unsigned cc = threadIdx.x >> 8, lt = threadIdx.x & 255u;
    const unsigned c = cr * 3 + cc, fc = f * C + c;
    const int y0 = b * R, nrows = min(R, H - y0);
    const unsigned nchunks = nrows * cpr;
    const u32x4* sb = s + ((size_t)fc * H + y0) * cpr;
    u32x4* db = d + ((size_t)f * 2 * H + (size_t)cr * H + y0) * 3 * cpr + (size_t)cc * cpr;
    for (unsigned base = lt; base < nchunks; base += 256 * U) {
        u32x4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) { unsigned i = base + j * 256; if (i < nchunks) v[j] = __builtin_nontemporal_load(sb + i); }
#pragma unroll
        for (int j = 0; j < U; ++j) { unsigned i = base + j * 256; if (i < nchunks) { unsigned r = i / cpr, col = i - r * cpr; __builtin_nontemporal_store(v[j], db + (size_t)r * 3 * cpr + col); } }
    }
}
// band split into TX column tiles, one 16-byte chunk per thread (U = 1), order (frame, camera row, band, camera column, tile)
__global__ __launch_bounds__(256) void k_copy_mosaic_tiles(const u32x4* __restrict__ s, u32x4* __restrict__ d, int C, int H, int cpr, int R, int NB, int TX)
{
    unsigned t = blockIdx.x;
    const unsigned tx = t % TX; t /= TX;
    const unsigned cc = t % 3; t /= 3;
    const unsigned b = t % NB; t /= NB;
    const unsigned cr = t % 2; const unsigned f = t / 2;
    const unsigned c = cr * 3 + cc, fc = f * C + c;
    const unsigned cpt = cpr / TX;
    const int y0 = b * R, nrows = min(R, H - y0);
    const unsigned r = threadIdx.x / cpt, col = threadIdx.x - r * cpt + tx * cpt;
    if ((int)r >= nrows) return;
    const u32x4* sb = s + ((size_t)fc * H + y0) * cpr;
    u32x4* db = d + ((size_t)f * 2 * H + (size_t)cr * H + y0) * 3 * cpr + (size_t)cc * cpr;
    u32x4 v = __builtin_nontemporal_load(sb + (size_t)r * cpr + col);
    __builtin_nontemporal_store(v, db + (size_t)r * 3 * cpr + col);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <typename F> void timeit(const char* name, size_t bytes, F launch)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
    printf("%-34s %8.3f ms  %7.1f GB/s (read+write)\n", name, ms, 2.0 * bytes / ms / 1e6);
}
int main()
{
    size_t bytes = (size_t)40 * 6 * 900 * 1600 * 3;   // one step's source frames (dst mosaic has the same size)
    size_t n = bytes / 16;
    u32x4 *s, *d; CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes));
    CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 2, bytes));
    timeit("oneshot U=1 nt", bytes, [&] { hipLaunchKernelGGL((k_copy<1, true>), dim3((n + 255) / 256), dim3(256), 0, 0, s, d, n); });
    timeit("oneshot U=4 nt", bytes, [&] { hipLaunchKernelGGL((k_copy<4, true>), dim3((n + 1023) / 1024), dim3(256), 0, 0, s, d, n); });
    timeit("oneshot U=8 nt", bytes, [&] { hipLaunchKernelGGL((k_copy<8, true>), dim3((n + 2047) / 2048), dim3(256), 0, 0, s, d, n); });
    timeit("oneshot U=4 plain", bytes, [&] { hipLaunchKernelGGL((k_copy<4, false>), dim3((n + 1023) / 1024), dim3(256), 0, 0, s, d, n); });
    timeit("persistent 2048wg U=4 nt", bytes, [&] { hipLaunchKernelGGL((k_copy_persist<4, true>), dim3(2048), dim3(256), 0, 0, s, d, n); });
    timeit("persistent 1024wg U=8 nt", bytes, [&] { hipLaunchKernelGGL((k_copy_persist<8, true>), dim3(1024), dim3(256), 0, 0, s, d, n); });
    timeit("persistent 4096wg U=2 nt", bytes, [&] { hipLaunchKernelGGL((k_copy_persist<2, true>), dim3(4096), dim3(256), 0, 0, s, d, n); });
    {
        int C = 6, H = 900, cpr = 300, F = 40;
        timeit("mosaic rows (1 chunk/thread)", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_rows<4>), dim3(F * C * H), dim3(320), 0, 0, s, d, C, H, cpr); });
        timeit("mosaic rows + 25KB LDS/WG", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_rows<25600>), dim3(F * C * H), dim3(320), 0, 0, s, d, C, H, cpr); });
        timeit("mosaic bands R=8 U=5 no LDS", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_bands<5, 4>), dim3(F * C * 113), dim3(256), 0, 0, s, d, C, H, cpr, 8, 113); });
        timeit("mosaic bands R=8 U=5 51KB LDS", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_bands<5, 51200>), dim3(F * C * 113), dim3(256), 0, 0, s, d, C, H, cpr, 8, 113); });
        timeit("mosaic bands R=4 U=5 no LDS", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_bands<5, 4>), dim3(F * C * 225), dim3(256), 0, 0, s, d, C, H, cpr, 4, 225); });
        timeit("mosaic bands R=4 U=5 25KB LDS", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_bands<5, 25600>), dim3(F * C * 225), dim3(256), 0, 0, s, d, C, H, cpr, 4, 225); });
        timeit("rowmajor bands R=8 U=5", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_bands_rowmajor<5>), dim3(F * C * 113), dim3(256), 0, 0, s, d, C, H, cpr, 8, 113); });
        timeit("rowmajor bands R=4 U=5", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_bands_rowmajor<5>), dim3(F * C * 225), dim3(256), 0, 0, s, d, C, H, cpr, 4, 225); });
        timeit("fullrow 768thr R=4 U=5", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_fullrow<5>), dim3(F * 2 * 225), dim3(768), 0, 0, s, d, C, H, cpr, 4, 225); });
        timeit("fullrow 768thr R=2 U=3", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_fullrow<3>), dim3(F * 2 * 450), dim3(768), 0, 0, s, d, C, H, cpr, 2, 450); });
        timeit("fullrow 768thr R=8 U=5", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_fullrow<5>), dim3(F * 2 * 113), dim3(768), 0, 0, s, d, C, H, cpr, 8, 113); });
        timeit("tiles TX=5 R=4 (240 thr, U=1)", bytes, [&] { hipLaunchKernelGGL(k_copy_mosaic_tiles, dim3(F * C * 225 * 5), dim3(256), 0, 0, s, d, C, H, cpr, 4, 225, 5); });
        timeit("tiles TX=3 R=2 (200 thr, U=1)", bytes, [&] { hipLaunchKernelGGL(k_copy_mosaic_tiles, dim3(F * C * 450 * 3), dim3(256), 0, 0, s, d, C, H, cpr, 2, 450, 3); });
        timeit("tiles TX=10 R=8 (240 thr, U=1)", bytes, [&] { hipLaunchKernelGGL(k_copy_mosaic_tiles, dim3(F * C * 113 * 10), dim3(256), 0, 0, s, d, C, H, cpr, 8, 113, 10); });
        timeit("rowmajor bands R=2 U=3", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_bands_rowmajor<3>), dim3(F * C * 450), dim3(256), 0, 0, s, d, C, H, cpr, 2, 450); });
        timeit("rowmajor bands R=1 U=2", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_bands_rowmajor<2>), dim3(F * C * 900), dim3(256), 0, 0, s, d, C, H, cpr, 1, 900); });
        timeit("rowmajor bands R=4 U=3", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_bands_rowmajor<3>), dim3(F * C * 225), dim3(256), 0, 0, s, d, C, H, cpr, 4, 225); });
        timeit("mosaic bands R=2 U=3 no LDS", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_bands<3, 4>), dim3(F * C * 450), dim3(256), 0, 0, s, d, C, H, cpr, 2, 450); });
        timeit("mosaic bands R=1 U=2 no LDS", bytes, [&] { hipLaunchKernelGGL((k_copy_mosaic_bands<2, 4>), dim3(F * C * 900), dim3(256), 0, 0, s, d, C, H, cpr, 1, 900); });
    }
    timeit("hipMemcpyDtoD", bytes, [&] { CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
