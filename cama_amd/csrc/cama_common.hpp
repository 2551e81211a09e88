// cama_common.hpp -- what the translation units of libcama_hip.so share on the HOST side (not part of the ABI):
// the thread-local error string behind cama_last_error(), the HIP error check, a rounding helper.
//   cama_hip.hip       kernels of the reprojection path + the single-stream entry points (the thin part of the contract)
//   cama_pipeline.hip  the pipeline runtime (cama_pipeline_*): streams, event rings, demand-sized scratch; no kernels
//   cama_jpeg.hip      device JPEG decoder (kernels + entry points) and the file-reader helper
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>

#include "cama_hip_diag.h"      // (includes cama_hip.h: the contract)

namespace cama_impl {
__attribute__((visibility("hidden"), format(printf, 2, 3))) int fail(int code, const char *fmt, ...);
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}  // namespace cama_impl

#define HIP_TRY(expr)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return cama_impl::fail(CAMA_EHIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)
