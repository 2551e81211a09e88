// Which engine does the runtime use for a device -> pinned-host copy, and how fast is it?  (Not part of the product.)
//   hipcc -O2 tools/ubench/d2h_engine.cpp -o tools/ubench/d2h_engine
//   d2h_engine <MB> <mode>   mode: async | dtoh | hostreg | wc | numa
// Run under `rocprofv3 --kernel-trace --memory-copy-trace --stats` to see whether the copy shows up as a blit kernel
// (__amd_rocclr_copyBuffer) or as an SDMA MEMORY_COPY.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 150) << 20;
    const char *mode = argc > 2 ? argv[2] : "async";
    void *d = nullptr, *h = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 7, bytes));
    if (!strcmp(mode, "hostreg")) {
        h = aligned_alloc(4096, bytes);
        memset(h, 0, bytes);
        CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
    } else if (!strcmp(mode, "wc")) CK(hipHostMalloc(&h, bytes, hipHostMallocWriteCombined));
    else if (!strcmp(mode, "numa")) CK(hipHostMalloc(&h, bytes, hipHostMallocNumaUser));
    else if (!strcmp(mode, "noncoh")) CK(hipHostMalloc(&h, bytes, hipHostMallocNonCoherent));
    else CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipStream_t s2;
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    void *d2 = nullptr;
    CK(hipMalloc(&d2, bytes));
    for (int rep = 0; rep < 2; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        const int n = 10;
        for (int k = 0; k < n; ++k) {
            if (!strcmp(mode, "afterkernel")) {                 // a kernel in the same stream right before the copy
                CK(hipMemsetAsync(d, k, 4096, s));
                CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
            } else if (!strcmp(mode, "otherstream")) {          // kernel on s, copy on s2 behind an event
                CK(hipMemsetAsync(d, k, 4096, s));
                CK(hipEventRecord(ev, s));
                CK(hipStreamWaitEvent(s2, ev, 0));
                CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s2));
            } else if (!strcmp(mode, "busyother")) {            // a long kernel running on s while s2 copies
                CK(hipMemsetAsync(d2, k, bytes, s));
                CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s2));
            } else
            if (!strcmp(mode, "dtoh")) CK(hipMemcpyDtoHAsync(h, (hipDeviceptr_t)d, bytes, s));
            else CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
        }
        CK(hipStreamSynchronize(s));
        CK(hipStreamSynchronize(s2));
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rep) printf("%s: %d x %zu MB down in %.2f ms = %.1f GB/s (first byte %d)\n", mode, n, bytes >> 20, sec * 1e3, n * bytes / sec / 1e9, ((unsigned char *)h)[0]);
    }
    return 0;
}
