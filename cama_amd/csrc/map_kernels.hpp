// map_kernels.hpp -- part of libcama_hip.so (included by cama_hip.hip inside its anonymous namespace).
// Per-clip static-map build kernel.
#pragma once

// ------------------------------------------------------------------------------------------
// static-map build (per clip): densify labels, lift with the BEV height raster, pixel -> world
// reproject.py:42-106.  One thread per OUTPUT point; float32 arithmetic in the reference's operation order
// (compiled -ffp-contract=off; HIP's float division is correctly rounded), so the buffer is bit-identical to the
// host build.  Writes the SoA vertex buffer + colour ids the fused render consumes: the map never visits the host.
// ------------------------------------------------------------------------------------------
struct MapBuildArgs {
    const float *verts;        // [V,2] label vertices (float32, as np.array(data).astype(np.float32))
    const int32_t *seg_v0;     // [S] first vertex of each non-empty segment (its end is v0 + 1)
    const int32_t *seg_num;    // [S] points emitted by the segment = int(|seg| / solution) > 0
    const int64_t *seg_off;    // [S+1] exclusive scan of seg_num
    const uint8_t *seg_colour; // [S]
    int32_t S;
    int64_t N;
    int32_t lift;              // 1: CAMA labels (BEV pixels + raster), 0: nuScenes labels (metres, z = 0)
    const void *raster;        // [rows, cols] float32 / float64
    int32_t rows, cols;
    float solution, half_w, half_h, cx, cy;
    void *x, *y, *z;           // [N] each, float32 or float64 (TZ)
    uint8_t *colour;           // [N]
};

template <typename TZ>
__global__ __launch_bounds__(BLOCK) void k_build_map(MapBuildArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.N) return;
    // segment that owns output point i: last s with seg_off[s] <= i
    int lo = 0, hi = a.S - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.seg_off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    const int s = lo;
    const float j = (float)(i - a.seg_off[s]);
    const float num = (float)a.seg_num[s];
    const float2 p0 = reinterpret_cast<const float2 *>(a.verts)[a.seg_v0[s]];
    const float2 p1 = reinterpret_cast<const float2 *>(a.verts)[a.seg_v0[s] + 1];
    // start + (end - start) / num * j      (reproject.py:62 / :92)
    const float px = p0.x + ((p1.x - p0.x) / num) * j;
    const float py = p0.y + ((p1.y - p0.y) / num) * j;
    TZ ox, oy, oz;
    if (a.lift) {
        // round().astype(np.uint16)[:, ::-1].clip(0, rows-1): half-to-even, C cast through int32 (wraps), (row, col)
        const int row = min(max((int)(uint16_t)(int32_t)rintf(py), 0), a.rows - 1);
        const int col = min(max((int)(uint16_t)(int32_t)rintf(px), 0), a.rows - 1);
        oz = static_cast<const TZ *>(a.raster)[(size_t)row * a.cols + col];
        // world x from pixel y and vice versa (reproject.py:38-39), float32
        ox = (TZ)(((py * a.solution) - a.half_w) + a.cx);
        oy = (TZ)(((px * a.solution) - a.half_h) + a.cy);
    } else {
        ox = (TZ)px; oy = (TZ)py; oz = (TZ)0;
    }
    static_cast<TZ *>(a.x)[i] = ox;
    static_cast<TZ *>(a.y)[i] = oy;
    static_cast<TZ *>(a.z)[i] = oz;
    a.colour[i] = a.seg_colour[s];
}
