"""Opportunistic pins against REAL OpenCV.

OpenCV is installed on neither box, so the three cv2 call sites of the hot path -- cv2.circle (cama/reproject.py:256),
cv2.initUndistortRectifyMap + cv2.remap (cama/reproject.py:238-239) -- are restated from OpenCV's published algorithms
(oracle/cama_oracle.{c,py}; "parity unpinned").  The day a box has cv2, these tests pin the restatements (and through
them the kernels, which are byte-equal to the restatements under -m gpu) to the real thing; until then they skip."""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

from oracle import cama_oracle as O          # noqa: E402

K0 = np.array([[1266.417203, 0.0, 816.2670197], [0.0, 1266.417203, 491.50706579], [0.0, 0.0, 1.0]])


def _disc_mask(radius, size, centre):
    hw = O.circle_halfwidths(radius)
    m = np.zeros((size, size), np.uint8)
    cy, cx = centre
    for dy in range(-radius, radius + 1):
        w = int(hw[abs(dy)])
        y = cy + dy
        if w < 0 or not 0 <= y < size:
            continue
        m[y, max(cx - w, 0):min(cx + w, size - 1) + 1] = 255
    return m


@pytest.mark.parametrize("radius", [0, 1, 2, 3, 4, 5])
def test_filled_circle_footprint(radius):
    size = 2 * radius + 5
    for centre in [(size // 2, size // 2), (0, 0), (1, size - 1), (size - 1, 2)]:       # centre, corners, edges: clipping
        img = np.zeros((size, size), np.uint8)
        cv2.circle(img, (centre[1], centre[0]), radius, 255, -1)
        assert np.array_equal(img, _disc_mask(radius, size, centre)), (radius, centre)


def test_filled_circle_c_oracle_and_library_table():
    """The C oracle's circle (what every -m gpu overlay test compares with) and the product library's half-width table."""
    img = np.zeros((7, 7, 3), np.uint8)
    cv2.circle(img, (3, 3), 2, (1, 2, 3), -1)
    mine = np.zeros((7, 7, 3), np.uint8)
    O.lib().oracle_circle_fill(O._ptr(mine), 7, 7, 21, 3, 3, 2, 1, 2, 3)
    assert np.array_equal(img, mine)
    assert int((img[..., 0] > 0).sum()) == 13                       # the 13-pixel diamond (rows 1/3/5/3/1)
    from cama_amd import _lib
    for r in range(0, 8):
        assert np.array_equal(_lib.circle_halfwidths(r), O.circle_halfwidths(r))


def _maps(W, H, dist):
    Kn = K0.copy()
    Kn[0] *= W / 1600
    Kn[1] *= H / 900
    d = np.asarray(dist, np.float64)
    mx, my = cv2.initUndistortRectifyMap(K0, d, None, Kn, (W, H), cv2.CV_32FC1)
    return Kn, mx, my


@pytest.mark.parametrize("dist", [[0.0] * 8, [-0.21, 0.07, 1e-3, -2e-3, 0.01, 0.02, -0.01, 0.003]])
def test_undistort_rectify_map(dist):
    """Float maps: the restatement follows OpenCV's scalar loop; OpenCV may run a SIMD body instead (different last-ulp
    rounding), so the float values are compared to 2^-12 px and the quantity cv2.remap consumes -- cvRound(map * 32) --
    exactly for the zero-distortion case (where it is provably stable) and to <= 1 step otherwise."""
    for W, H in [(960, 540), (48, 27)]:
        Kn, mx, my = _maps(W, H, dist)
        ox, oy = O.undistort_map(K0, dist, Kn, W, H)
        assert np.abs(mx - ox).max() <= 2.0 ** -12 and np.abs(my - oy).max() <= 2.0 ** -12
        qx, qy = np.rint(mx * np.float32(32)), np.rint(my * np.float32(32))
        rx, ry = np.rint(ox * np.float32(32)), np.rint(oy * np.float32(32))
        if not any(dist):
            assert np.array_equal(qx, rx) and np.array_equal(qy, ry)
        else:
            assert np.abs(qx - rx).max() <= 1 and np.abs(qy - ry).max() <= 1
        print("map exact-equal fraction", float((mx == ox).mean()), float((my == oy).mean()))


def test_remap_bilinear_45x80_to_27x48():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (45, 80, 3), dtype=np.uint8)
    Kn = K0.copy()
    Kn[0] *= 48 / 1600
    Kn[1] *= 27 / 900
    Ks = K0.copy()                      # a 45x80 "sensor": scale the origin calibration down too
    Ks[0] *= 80 / 1600
    Ks[1] *= 45 / 900
    for dist in ([0.0] * 8, [-0.3, 0.1, 1e-3, -2e-3, 0.0, 0.0, 0.0, 0.0]):
        mx, my = cv2.initUndistortRectifyMap(Ks, np.asarray(dist), None, Kn, (48, 27), cv2.CV_32FC1)
        want = cv2.remap(img, mx, my, cv2.INTER_LINEAR)
        assert np.array_equal(want, O.remap_bilinear(img, mx, my))
    # maps that leave the image on every side: constant border 0
    jj, ii = np.meshgrid(np.arange(48, dtype=np.float32), np.arange(27, dtype=np.float32))
    mx, my = jj * 2.1 - 7.3, ii * 2.2 - 5.9
    assert np.array_equal(cv2.remap(img, mx, my, cv2.INTER_LINEAR), O.remap_bilinear(img, mx, my))


def test_reference_render_maps_loop_on_real_cv2():
    """cama/reproject.py:246-257 with real cv2.circle against the oracle's render of the same instances."""
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (90, 160, 3), dtype=np.uint8)
    maps_2d = [{"class": "lane_marking", "points": np.stack([rng.uniform(0, 90, 50), rng.uniform(0, 160, 50)], -1)},
               {"class": "Road_teeth", "points": np.stack([rng.uniform(0, 90, 70), rng.uniform(0, 160, 70)], -1)}]
    want = img.copy()
    for ins in maps_2d:
        pts = ins["points"].astype(np.int32)
        bgr = O.colour_bgr(ins["class"])
        for v, u in pts:
            cv2.circle(want, (int(u), int(v)), 2, tuple(int(c) for c in bgr), -1)
    got = O.render_instances(img.copy(), maps_2d)
    assert np.array_equal(got, want)
