/*
 * oracle/cama_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the per-frame arithmetic of the CAMA
 * reprojection hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (cama_amd/) never does.
 *
 * What is restated, and from where:
 *   oracle_frame_project  <- cama/reproject.py:108-116 (homogeneous transform),
 *                            :118-131 (inclusive crop box), :187-205 (K, z>0,
 *                            divide, in-image mask, (v,u) output order);
 *                            cama/dataset.py:101-112 (world->chassis once, then
 *                            chassis->camera per camera on the CROPPED points).
 *   oracle_circle_fill    <- cv2.circle(img,(u,v),r,color,-1) as called at
 *                            cama/reproject.py:256.  OpenCV is an un-vendored,
 *                            unpinned dependency (requirements.txt:5) that is
 *                            not installed on either box: this is a restatement
 *                            of its published integer midpoint algorithm
 *                            (imgproc/drawing.cpp, static Circle(), fill branch,
 *                            taken for thickness<0, LINE_8, shift 0).
 *                            PARITY UNPINNED for the footprint itself.
 *   oracle_render_frame   <- cama/reproject.py:246-257 (truncate to int32,
 *                            sequential draw, last writer wins) + tools.py:22-25
 *                            (2x3 mosaic).
 *
 * Arithmetic contract (pinned against the golden vectors, tests/golden/):
 * numpy's float64 matmul of a (4,4)/(3,3) matrix with a (4,n)/(3,n) block is,
 * for n >= 2, a k-ordered FMA chain  acc = m0*p0; acc = fma(m1,p1,acc); ...
 * (measured in this container against OpenBLAS 0.3.29: 0 mismatches in 4e5
 * outputs; separate mul/add mismatches in ~30 %).  This file states that chain
 * explicitly with fma(), so it is deterministic on any host; the HIP kernels
 * state the same chain with __builtin_fma.  Single-point instances (n == 1) go
 * through BLAS gemv in the reference and may differ by 1 ulp (documented).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

/* ---- 3x4 affine rows of a 4x4 row-major matrix applied to (x,y,z,1) ---- */
static inline void xform_rows(const double *T, double x, double y, double z,
                              double *ox, double *oy, double *oz)
{
    double a;
    a = T[0] * x;  a = fma(T[1], y, a);  a = fma(T[2], z, a);  a = fma(T[3], 1.0, a);  *ox = a;
    a = T[4] * x;  a = fma(T[5], y, a);  a = fma(T[6], z, a);  a = fma(T[7], 1.0, a);  *oy = a;
    a = T[8] * x;  a = fma(T[9], y, a);  a = fma(T[10], z, a); a = fma(T[11], 1.0, a); *oz = a;
}

static inline void k_rows(const double *K, double x, double y, double z,
                          double *o0, double *o1, double *o2)
{
    double a;
    a = K[0] * x; a = fma(K[1], y, a); a = fma(K[2], z, a); *o0 = a;
    a = K[3] * x; a = fma(K[4], y, a); a = fma(K[5], z, a); *o1 = a;
    a = K[6] * x; a = fma(K[7], y, a); a = fma(K[8], z, a); *o2 = a;
}

/*
 * One frame.  xyz: N x 3 map-frame points, float (is_f64 == 0) or double.
 * w2c: 4x4 row-major doubles (the float32 inverse, promoted exactly).
 * c2cam: C x 16, K: C x 9 doubles.  crop: xmin,xmax,ymin,ymax,zmin,zmax.
 * Outputs (any may be NULL): chassis N x 3, crop_mask N,
 * vu C x N x 2 (v first), vis C x N.  Entries of vu where vis==0 are left
 * untouched except that they are written when the point passed the crop
 * (values may be inf/nan exactly like the reference's intermediate).
 */
void oracle_frame_project(const void *xyz, int is_f64, int64_t N,
                          const double *w2c, const double *crop,
                          const double *c2cam, const double *K, int C,
                          int W, int H,
                          double *chassis, uint8_t *crop_mask,
                          double *vu, uint8_t *vis)
{
    for (int64_t i = 0; i < N; i++) {
        double x, y, z;
        if (is_f64) {
            const double *p = (const double *)xyz + 3 * i;
            x = p[0]; y = p[1]; z = p[2];
        } else {
            const float *p = (const float *)xyz + 3 * i;
            x = p[0]; y = p[1]; z = p[2];
        }
        double cx, cy, cz;
        xform_rows(w2c, x, y, z, &cx, &cy, &cz);
        if (chassis) { chassis[3 * i] = cx; chassis[3 * i + 1] = cy; chassis[3 * i + 2] = cz; }
        int in = (cx >= crop[0]) & (cx <= crop[1]) & (cy >= crop[2]) & (cy <= crop[3]) &
                 (cz >= crop[4]) & (cz <= crop[5]);
        if (crop_mask) crop_mask[i] = (uint8_t)in;
        for (int c = 0; c < C; c++) {
            uint8_t ok = 0;
            if (in) {
                double px, py, pz, h0, h1, h2;
                xform_rows(c2cam + 16 * c, cx, cy, cz, &px, &py, &pz);
                k_rows(K + 9 * c, px, py, pz, &h0, &h1, &h2);
                double u = h0 / h2, v = h1 / h2, w = h2 / h2;
                ok = (uint8_t)((h2 > 0) & (w > 0) & (u >= 0) & (u < (double)W) & (v >= 0) & (v < (double)H));
                if (vu) { vu[((int64_t)c * N + i) * 2] = v; vu[((int64_t)c * N + i) * 2 + 1] = u; }
            }
            if (vis) vis[(int64_t)c * N + i] = ok;
        }
    }
}

/* Generic single-transform helper (reproject.py:108-116) for API-level checks. */
void oracle_transform(const double *xyz, int64_t N, const double *T, double *out)
{
    for (int64_t i = 0; i < N; i++)
        xform_rows(T, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], out + 3 * i, out + 3 * i + 1, out + 3 * i + 2);
}

/* ---- OpenCV-style filled circle on an H x W x 3 uint8 image with row stride `step` bytes ---- */
static inline void hline(uint8_t *row, int x0, int x1, const uint8_t *bgr)
{
    for (int x = x0; x <= x1; x++) { row[3 * x] = bgr[0]; row[3 * x + 1] = bgr[1]; row[3 * x + 2] = bgr[2]; }
}

void oracle_circle_fill(uint8_t *img, int H, int W, int64_t step, int cx, int cy, int radius,
                        int b, int g, int r)
{
    const uint8_t bgr[3] = {(uint8_t)b, (uint8_t)g, (uint8_t)r};
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    int inside = cx >= radius && cx < W - radius && cy >= radius && cy < H - radius;
    while (dx >= dy) {
        int mask;
        int y11 = cy - dy, y12 = cy + dy, y21 = cy - dx, y22 = cy + dx;
        int x11 = cx - dx, x12 = cx + dx, x21 = cx - dy, x22 = cx + dy;
        if (inside) {
            hline(img + (int64_t)y11 * step, x11, x12, bgr);
            hline(img + (int64_t)y12 * step, x11, x12, bgr);
            hline(img + (int64_t)y21 * step, x21, x22, bgr);
            hline(img + (int64_t)y22 * step, x21, x22, bgr);
        } else if (x11 < W && x12 >= 0 && y21 < H && y22 >= 0) {
            if (x11 < 0) x11 = 0;
            if (x12 > W - 1) x12 = W - 1;
            if ((unsigned)y11 < (unsigned)H) hline(img + (int64_t)y11 * step, x11, x12, bgr);
            if ((unsigned)y12 < (unsigned)H) hline(img + (int64_t)y12 * step, x11, x12, bgr);
            if (x21 < W && x22 >= 0) {
                if (x21 < 0) x21 = 0;
                if (x22 > W - 1) x22 = W - 1;
                if ((unsigned)y21 < (unsigned)H) hline(img + (int64_t)y21 * step, x21, x22, bgr);
                if ((unsigned)y22 < (unsigned)H) hline(img + (int64_t)y22 * step, x21, x22, bgr);
            }
        }
        dy++;
        err += plus;
        plus += 2;
        mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

/* EXTENSION restatement (no reference counterpart: the reference draws discs only).  One-pixel-wide, 8-connected integer
 * Bresenham segment from (x0,y0) to (x1,y1), both end pixels included, every octant and direction; pixels outside the image
 * are skipped.  This recurrence IS the definition the HIP kernel (k_segments_global) is checked against. */
void oracle_line_bresenham(uint8_t *img, int H, int W, int64_t step, int x0, int y0, int x1, int y1, int b, int g, int r)
{
    const int dx = abs(x1 - x0), sx = x0 < x1 ? 1 : -1;
    const int dy = -abs(y1 - y0), sy = y0 < y1 ? 1 : -1;
    int err = dx + dy;
    for (;;) {
        if ((unsigned)x0 < (unsigned)W && (unsigned)y0 < (unsigned)H) {
            uint8_t *p = img + (int64_t)y0 * step + 3 * x0;
            p[0] = (uint8_t)b; p[1] = (uint8_t)g; p[2] = (uint8_t)r;
        }
        if (x0 == x1 && y0 == y1) break;
        const int e2 = 2 * err;
        if (e2 >= dy) { err += dy; x0 += sx; }
        if (e2 <= dx) { err += dx; y0 += sy; }
    }
}

/* Per-row half-widths of the union footprint of the algorithm above, hw[0..radius];
 * row offsets +-k get half-width hw[k]; returns radius+1.  Used by tests to derive
 * the table the HIP overlay kernel takes as data. */
int oracle_circle_halfwidths(int radius, int *hw)
{
    for (int k = 0; k <= radius; k++) hw[k] = -1;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        if (dx > hw[dy]) hw[dy] = dx;
        if (dy > hw[dx]) hw[dx] = dy;
        dy++; err += plus; plus += 2;
        int mask = (err <= 0) - 1;
        err -= minus & mask; dx += mask; minus -= mask & 2;
    }
    return radius + 1;
}

/*
 * Render one frame: for every camera copy src[c] into the mosaic cell
 * (tools.py:22-25: row = c/3, col = c%3 for the fixed 6-camera order) and then
 * draw the visible points in index order (reproject.py:246-257).
 * vu: C x N x 2 doubles (v,u), vis: C x N, colour_id: N (0 = lane grey, 1 = gold),
 * palette_bgr: 2 x 3.  src: C x H x W x 3.  mosaic: (rows*H) x (cols*W) x 3.
 */
void oracle_render_frame(const uint8_t *src, uint8_t *mosaic, int C, int H, int W,
                         int cols, const double *vu, const uint8_t *vis,
                         const uint8_t *colour_id, int64_t N, int radius,
                         const uint8_t *palette_bgr)
{
    int64_t step = (int64_t)cols * W * 3;
    for (int c = 0; c < C; c++) {
        uint8_t *cell = mosaic + (int64_t)(c / cols) * H * step + (int64_t)(c % cols) * W * 3;
        for (int y = 0; y < H; y++)
            memcpy(cell + (int64_t)y * step, src + (((int64_t)c * H + y) * W) * 3, (size_t)W * 3);
        for (int64_t i = 0; i < N; i++) {
            if (!vis[(int64_t)c * N + i]) continue;
            int32_t vi = (int32_t)vu[((int64_t)c * N + i) * 2];      /* astype(np.int32): truncation */
            int32_t ui = (int32_t)vu[((int64_t)c * N + i) * 2 + 1];
            const uint8_t *p = palette_bgr + 3 * colour_id[i];
            oracle_circle_fill(cell, H, W, step, ui, vi, radius, p[0], p[1], p[2]);
        }
    }
}

/*
 * EXTENSION checker (no reference semantics: the reference is opaque): translucent stamps.  A pixel covered by at
 * least one disc becomes (colour*alpha256 + source*(256-alpha256) + 128) >> 8 per byte, colour = the last writer's.
 * Implemented as: draw the discs into a colour layer pre-filled with a sentinel, then composite once.
 */
void oracle_render_frame_alpha(const uint8_t *src, uint8_t *mosaic, int C, int H, int W, int cols, const double *vu,
                               const uint8_t *vis, const uint8_t *colour_id, int64_t N, int radius,
                               const uint8_t *palette_bgr, int alpha256, uint8_t *layer /* H*W*3 scratch */)
{
    int64_t step = (int64_t)cols * W * 3;
    const uint8_t sentinel[3] = {1, 2, 3};
    for (int c = 0; c < C; c++) {
        uint8_t *cell = mosaic + (int64_t)(c / cols) * H * step + (int64_t)(c % cols) * W * 3;
        for (int64_t p = 0; p < (int64_t)H * W; p++) memcpy(layer + 3 * p, sentinel, 3);
        for (int64_t i = 0; i < N; i++) {
            if (!vis[(int64_t)c * N + i]) continue;
            int32_t vi = (int32_t)vu[((int64_t)c * N + i) * 2], ui = (int32_t)vu[((int64_t)c * N + i) * 2 + 1];
            const uint8_t *q = palette_bgr + 3 * colour_id[i];
            oracle_circle_fill(layer, H, W, (int64_t)W * 3, ui, vi, radius, q[0], q[1], q[2]);
        }
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const uint8_t *s3 = src + (((int64_t)c * H + y) * W + x) * 3, *l3 = layer + ((int64_t)y * W + x) * 3;
                uint8_t *d3 = cell + (int64_t)y * step + (int64_t)x * 3;
                int owned = memcmp(l3, sentinel, 3) != 0;
                for (int k = 0; k < 3; k++)
                    d3[k] = owned ? (uint8_t)((l3[k] * alpha256 + s3[k] * (256 - alpha256) + 128) >> 8) : s3[k];
            }
    }
}

/*
 * EXTENSION restatement (no reference counterpart; the north-star's "Wu line-raster + blend"): discs plus ANTI-ALIASED
 * one-pixel segments between polyline neighbours.  Definition (this code IS the definition the HIP kernel is checked against):
 *   every visible point k, in draw order, claims (a) the pixels of its disc with coverage 255 and (b) -- when link[k] is set,
 *   point k - 1 is visible in the camera too and lies on another pixel -- the pixels of the Wu line from point k - 1's pixel
 *   to its own with a coverage in 1..255; a pixel shows the claim with the greatest (k, coverage) and is blended ONCE over the
 *   source: (colour * a + source * (256 - a) + 128) >> 8 with a = coverage + (coverage >> 7)  (255 -> 256: opaque).
 * Wu's line between integer pixel centres, major axis stepped pixel by pixel, minor coordinate in 16.16 fixed point:
 *   gradient = floor((d_minor << 16) / d_major); at step j: y = (minor0 << 16) + gradient * j; row = y >> 16 (floor),
 *   f = (y & 0xffff) >> 8; (major0 + j, row) gets 255 - f and (major0 + j, row + 1) gets f; zero coverages claim nothing.
 * claim[] = H*W uint32 scratch, ((k + 1) << 8) | coverage; k < 2^23.
 */
static void wu_claim(uint32_t *claim, int H, int W, int x, int y, uint32_t v)
{
    if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H && (v & 255u) && claim[(int64_t)y * W + x] < v)
        claim[(int64_t)y * W + x] = v;
}

static void wu_line_claims(uint32_t *claim, int H, int W, int x0, int y0, int x1, int y1, uint32_t key1)
{
    const int steep = abs(y1 - y0) > abs(x1 - x0);
    int a0 = steep ? y0 : x0, b0 = steep ? x0 : y0, a1 = steep ? y1 : x1, b1 = steep ? x1 : y1;     /* a = major, b = minor */
    if (a0 > a1) { int t = a0; a0 = a1; a1 = t; t = b0; b0 = b1; b1 = t; }
    const int da = a1 - a0, db = b1 - b0;
    if (da == 0) {
        wu_claim(claim, H, W, x0, y0, (key1 << 8) | 255u);
        return;
    }
    int32_t num = db * 65536, grad = num / da;
    if (num % da != 0 && num < 0) grad -= 1;                                                    /* floor division (da > 0) */
    for (int j = 0; j <= da; j++) {
        const int32_t y = b0 * 65536 + grad * j;
        const int row = (int)(y >> 16);                         /* arithmetic shift: floor */
        const uint32_t f = ((uint32_t)y & 0xffffu) >> 8;
        const int a = a0 + j;
        if (steep) {
            wu_claim(claim, H, W, row, a, (key1 << 8) | (255u - f));
            wu_claim(claim, H, W, row + 1, a, (key1 << 8) | f);
        } else {
            wu_claim(claim, H, W, a, row, (key1 << 8) | (255u - f));
            wu_claim(claim, H, W, a, row + 1, (key1 << 8) | f);
        }
    }
}

void oracle_render_frame_wu(const uint8_t *src, uint8_t *mosaic, int C, int H, int W, int cols, const double *vu,
                            const uint8_t *vis, const uint8_t *colour_id, const uint8_t *link, int64_t N, int radius,
                            const uint8_t *palette_bgr, uint32_t *claim /* H*W scratch */)
{
    const int64_t step = (int64_t)cols * W * 3;
    int hw[64];
    oracle_circle_halfwidths(radius, hw);
    for (int c = 0; c < C; c++) {
        uint8_t *cell = mosaic + (int64_t)(c / cols) * H * step + (int64_t)(c % cols) * W * 3;
        memset(claim, 0, (size_t)H * W * 4);
        for (int64_t i = 0; i < N; i++) {
            if (!vis[(int64_t)c * N + i]) continue;
            const int vi = (int32_t)vu[((int64_t)c * N + i) * 2], ui = (int32_t)vu[((int64_t)c * N + i) * 2 + 1];
            const uint32_t key1 = (uint32_t)i + 1u;
            if (i > 0 && link[i] && vis[(int64_t)c * N + i - 1]) {
                const int vp = (int32_t)vu[((int64_t)c * N + i - 1) * 2], up = (int32_t)vu[((int64_t)c * N + i - 1) * 2 + 1];
                if (vp != vi || up != ui) wu_line_claims(claim, H, W, up, vp, ui, vi, key1);
            }
            for (int dy = -radius; dy <= radius; dy++)          /* the disc: the footprint of oracle_circle_fill */
                for (int dx = -hw[abs(dy)]; dx <= hw[abs(dy)]; dx++) wu_claim(claim, H, W, ui + dx, vi + dy, (key1 << 8) | 255u);
        }
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const uint8_t *s3 = src + (((int64_t)c * H + y) * W + x) * 3;
                uint8_t *d3 = cell + (int64_t)y * step + (int64_t)x * 3;
                const uint32_t v = claim[(int64_t)y * W + x];
                if (!v) { d3[0] = s3[0]; d3[1] = s3[1]; d3[2] = s3[2]; continue; }
                const uint32_t cov = v & 255u, a = cov + (cov >> 7);
                const uint8_t *q = palette_bgr + 3 * colour_id[(v >> 8) - 1u];
                for (int k = 0; k < 3; k++) d3[k] = (uint8_t)((q[k] * a + s3[k] * (256u - a) + 128u) >> 8);
            }
    }
}

/*
 * cv2.initUndistortRectifyMap(K_origin, dist, None, K_new, (W,H), CV_32FC1) as called at
 * cama/reproject.py:238 -- restated from OpenCV's published source (imgproc/src/undistort.dispatch.cpp,
 * cv::initUndistortRectifyMap, and the scalar loop of initUndistortRectifyMapComputer in undistort.simd.hpp),
 * operation by operation, in double, stored as float:
 *
 *   iR = (K_new * R).inv(DECOMP_LU), R = identity.  cv::invert takes its closed-form 3x3 branch for that method:
 *        d = 1/det3(A); inverse = adjugate entries, each (a*b - c*d) * d                       [lapack.cpp, n == 3]
 *   per row i:   _x = i*ir[1] + ir[2], _y = i*ir[4] + ir[5], _w = i*ir[7] + ir[8]
 *   per column:  w = 1./_w; x = _x*w; y = _y*w;  ... distortion ...;  u = fx*invProj*xd' + u0;  v = fy*invProj*yd' + v0
 *                then _x += ir[0], _y += ir[3], _w += ir[6]      (ROW-INCREMENTAL: rounding accumulates along the row)
 *   u uses only fx and u0 of K_origin (no skew term).
 * Tilt (tauX, tauY = dist[12], dist[13]) is the identity for tau = 0; non-zero tilt is refused (returns -1): the
 * reference's calibrations carry 8 zero coefficients (dataset/nuscenes2clip.py:510-522).
 *
 * What this cannot pin (OpenCV is absent on both boxes, PARITY UNPINNED): OpenCV builds dispatch the row loop to a
 * SIMD body where the CPU has one (universal intrinsics / AVX2: lanes computed as (_x + k*ir[0]) * (1/(_w + k*ir[6])),
 * FMA where available), which rounds differently in the last ulp.  For the reference's zero-distortion 0.6 scale the
 * quantity the remap consumes, cvRound(map*32), is 160*j/3 -- its fraction is 0, 1/3 or 2/3, never near 1/2 -- so
 * last-ulp differences cannot change a pixel; tests/test_cv2_pins.py checks the real thing wherever cv2 exists.
 */
static double det3(const double *m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

int oracle_undistort_map(const double *K_origin, const double *dist, int ndist, const double *K_new, int W, int H,
                         float *mapx, float *mapy)
{
    double k[14] = {0};
    for (int i = 0; i < ndist && i < 14; ++i) k[i] = dist[i];
    if (k[12] != 0.0 || k[13] != 0.0) return -1;
    const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    const double s1 = k[8], s2 = k[9], s3 = k[10], s4 = k[11];
    const double *S = K_new;                         /* Ar.colRange(0,3) * R with R = eye: exact */
    double d = det3(S);
    if (d == 0.0) return -2;
    d = 1. / d;
    double ir[9];
    ir[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    ir[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    ir[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    ir[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    ir[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    ir[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    ir[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    ir[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    ir[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    const double u0 = K_origin[2], v0 = K_origin[5], fx = K_origin[0], fy = K_origin[4];
    for (int i = 0; i < H; ++i) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < W; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
            double w = 1. / _w, x = _x * w, y = _y * w;
            double x2 = x * x, y2 = y * y;
            double r2 = x2 + y2, _2xy = 2 * x * y;
            double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
            double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
            /* matTilt = identity: vecTilt = (xd, yd, 1), invProj = 1./1 */
            double invProj = 1.;
            double u = fx * invProj * xd + u0;
            double v = fy * invProj * yd + v0;
            mapx[(size_t)i * W + j] = (float)u;
            mapy[(size_t)i * W + j] = (float)v;
        }
    }
    return 0;
}
