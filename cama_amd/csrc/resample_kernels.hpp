// resample_kernels.hpp -- part of libcama_hip.so (included by cama_hip.hip inside its anonymous namespace).
// Stand-alone undistort + resize kernels.
#pragma once

// ------------------------------------------------------------------------------------------
// frame resample: cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) over float32 maps (reproject.py:238-239)
// OpenCV's 8-bit remap quantises coordinates to 1/32 px (INTER_BITS = 5) and blends with 15-bit fixed-point
// weights; for bilinear the weights (32-a)(32-b)*32 ... are exact integers summing to 1 << 15.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_resample(const uint8_t *__restrict__ src, int64_t src_stride,
                                                    uint8_t *__restrict__ dst, int64_t dst_stride, int H0, int W0,
                                                    int H, int W, const float *__restrict__ mapx,
                                                    const float *__restrict__ mapy, MapStride ms)
{
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= H * W) return;
    const uint8_t *s = src + (size_t)blockIdx.y * src_stride;
    uint8_t *d = dst + (size_t)blockIdx.y * dst_stride + (size_t)p * 3;
    const int y = (int)__umulhi((uint32_t)p, ms.w_magic), x = p - y * W;
    const uint32_t c = remap_pixel(s, (size_t)H0 * W0 * 3, H0, W0, mapx[y * ms.xr + x * ms.xc],
                                   mapy[y * ms.yr + x * ms.yc]);
    d[0] = (uint8_t)c; d[1] = (uint8_t)(c >> 8); d[2] = (uint8_t)(c >> 16);
}

// W % 16 == 0: thread <-> destination pixel for the gathers (adjacent lanes read adjacent source pixels, the map
// loads are coalesced), PPT pixels per thread strided by the workgroup size so the dependent load chains
// (map -> taps) of several pixels overlap; then the workgroup's bytes are transposed through LDS into aligned
// 16-byte stores.  (One thread per 16 CONSECUTIVE pixels was tried: lanes 80 source bytes apart lose all
// coalescing -- 4.7x slower; one pixel per thread is latency-bound at 8 workgroups/CU.)
#ifndef RESAMPLE_PPT_N
#define RESAMPLE_PPT_N 4
#endif
constexpr int RESAMPLE_PPT = RESAMPLE_PPT_N;

__global__ __launch_bounds__(BLOCK) void k_resample16(const uint8_t *__restrict__ src, int64_t src_stride,
                                                      uint8_t *__restrict__ dst, int64_t dst_stride, int H0, int W0,
                                                      int H, int W, const float *__restrict__ mapx,
                                                      const float *__restrict__ mapy, MapStride ms)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_out[BLOCK * RESAMPLE_PPT * 3];
    const int npix = H * W;
    const int first = blockIdx.x * (BLOCK * RESAMPLE_PPT);
    const uint8_t *s = src + (size_t)blockIdx.y * src_stride;
    float mx[RESAMPLE_PPT], my[RESAMPLE_PPT];
#pragma unroll
    for (int j = 0; j < RESAMPLE_PPT; ++j) {
        const int p = min(first + j * BLOCK + (int)threadIdx.x, npix - 1);
        const int y = (int)__umulhi((uint32_t)p, ms.w_magic), x = p - y * W;
        mx[j] = mapx[y * ms.xr + x * ms.xc];
        my[j] = mapy[y * ms.yr + x * ms.yc];
    }
#pragma unroll
    for (int j = 0; j < RESAMPLE_PPT; ++j) {
        const uint32_t c = remap_pixel(s, (size_t)H0 * W0 * 3, H0, W0, mx[j], my[j]);
        uint8_t *o = s_out + 3 * (j * BLOCK + (int)threadIdx.x);
        o[0] = (uint8_t)c; o[1] = (uint8_t)(c >> 8); o[2] = (uint8_t)(c >> 16);
    }
    __syncthreads();
    // npix is a multiple of 16, so the tail workgroup ends on a chunk boundary
    const int valid_chunks = (min(BLOCK * RESAMPLE_PPT, npix - first) * 3) >> 4;
    u32x4 *d = reinterpret_cast<u32x4 *>(dst + (size_t)blockIdx.y * dst_stride + (size_t)first * 3);
    for (int k = threadIdx.x; k < valid_chunks; k += BLOCK) d[k] = reinterpret_cast<const u32x4 *>(s_out)[k];
}
