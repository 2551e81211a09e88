// egress_kernels.hpp -- part of libcama_hip.so (included by cama_hip.hip inside its anonymous namespace).
// Mosaic egress: BGR24 -> planar YUV 4:2:0 (I420) on the device, so that what crosses PCIe and the encoder's pipe is
// 1.5 bytes per pixel instead of 3 and libx264's front end has no colour conversion left to do.  The reference pipes
// raw bgr24 into ffmpeg (cama/tools.py:13-20,27-32) and lets libswscale convert to yuv420p; this restates
// libswscale's unscaled BGR24 -> YUV420P C path (rgb2rgb_template.c, rgb24toyv12_c, called with the BGR byte order and
// the BT.601 limited-range table RY..BV, 15-bit fixed point): Y = ((ry*r + gy*g + by*b) >> 15) + 16 for every pixel,
// U/V = ((ru*r + gu*g + bu*b) >> 15) + 128 taken from the first pixel of the first line of each 2x2 block.
// PARITY UNPINNED: no ffmpeg on either box, and x86 builds of libswscale may run a SIMD body that averages chroma.
#pragma once

struct I420Coeffs { int ry, gy, by, ru, gu, bu, rv, gv, bv; };
__device__ constexpr I420Coeffs kI420 = {8414, 16519, 3208, -4865, -9528, 14392, 14392, -12061, -2332};

__device__ __forceinline__ uint32_t i420_y(uint32_t b, uint32_t g, uint32_t r)
{
    return (uint32_t)(((kI420.ry * (int)r + kI420.gy * (int)g + kI420.by * (int)b) >> 15) + 16);
}

// one thread = 16 pixels x 2 lines: 6 aligned 16-byte loads, two 16-byte Y stores, one 8-byte U and V store
__global__ __launch_bounds__(BLOCK) void k_bgr_to_i420(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int H, int W,
                                                       size_t src_stride, size_t dst_stride)
{
    const int bx = W >> 4, by = H >> 1;
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= bx * by) return;
    const int ty = t / bx, tx = t - ty * bx;
    const uint8_t *s = src + (size_t)blockIdx.y * src_stride + ((size_t)(2 * ty) * W + (size_t)tx * 16) * 3;
    uint8_t *d = dst + (size_t)blockIdx.y * dst_stride;
    uint8_t *dy = d + (size_t)(2 * ty) * W + (size_t)tx * 16;
    uint8_t *du = d + (size_t)H * W + (size_t)ty * (W >> 1) + (size_t)tx * 8;
    uint8_t *dv = du + (size_t)(H >> 1) * (W >> 1);
    uint32_t in0[12], in1[12];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const u32x4 a = reinterpret_cast<const u32x4 *>(s)[k];
        const u32x4 b = reinterpret_cast<const u32x4 *>(s + (size_t)W * 3)[k];
        in0[4 * k] = a.x; in0[4 * k + 1] = a.y; in0[4 * k + 2] = a.z; in0[4 * k + 3] = a.w;
        in1[4 * k] = b.x; in1[4 * k + 1] = b.y; in1[4 * k + 2] = b.z; in1[4 * k + 3] = b.w;
    }
    auto byte_of = [](const uint32_t *w, int i) -> uint32_t { return (w[i >> 2] >> (8 * (i & 3))) & 0xffu; };
    uint32_t y0[4] = {0, 0, 0, 0}, y1[4] = {0, 0, 0, 0}, u[2] = {0, 0}, v[2] = {0, 0};
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const uint32_t b0 = byte_of(in0, 3 * p), g0 = byte_of(in0, 3 * p + 1), r0 = byte_of(in0, 3 * p + 2);
        const uint32_t b1 = byte_of(in1, 3 * p), g1 = byte_of(in1, 3 * p + 1), r1 = byte_of(in1, 3 * p + 2);
        y0[p >> 2] |= i420_y(b0, g0, r0) << (8 * (p & 3));
        y1[p >> 2] |= i420_y(b1, g1, r1) << (8 * (p & 3));
        if ((p & 1) == 0) {
            const uint32_t U = (uint32_t)(((kI420.ru * (int)r0 + kI420.gu * (int)g0 + kI420.bu * (int)b0) >> 15) + 128);
            const uint32_t V = (uint32_t)(((kI420.rv * (int)r0 + kI420.gv * (int)g0 + kI420.bv * (int)b0) >> 15) + 128);
            u[p >> 3] |= (U & 0xffu) << (8 * ((p >> 1) & 3));
            v[p >> 3] |= (V & 0xffu) << (8 * ((p >> 1) & 3));
        }
    }
    u32x4 o;
    o.x = y0[0]; o.y = y0[1]; o.z = y0[2]; o.w = y0[3];
    *reinterpret_cast<u32x4 *>(dy) = o;
    o.x = y1[0]; o.y = y1[1]; o.z = y1[2]; o.w = y1[3];
    *reinterpret_cast<u32x4 *>(dy + W) = o;
    *reinterpret_cast<uint2 *>(du) = make_uint2(u[0], u[1]);
    *reinterpret_cast<uint2 *>(dv) = make_uint2(v[0], v[1]);
}
