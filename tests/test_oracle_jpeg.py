"""Pins oracle/jpeg_oracle.py (the CPU restatement of libjpeg-turbo's baseline decode the device JPEG decoder is
checked against) to the real decoder: Pillow's bundled libjpeg-turbo, byte for byte, across sizes (incl. non-multiples
of the MCU), chroma subsamplings, qualities, grayscale, restart intervals and optimised Huffman tables."""
import io

import numpy as np
import pytest
from PIL import Image

from oracle import jpeg_oracle as J


def synth_image(h, w, kind, seed=0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    if kind == "smooth":
        return np.stack([(x * 2.1 + 20 * np.sin(y / 7)) % 256, (y * 3.3) % 256, ((x + y) * 1.7) % 256], -1).astype(np.uint8)
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img = np.zeros((h, w, 3), np.uint8)
    img[h // 4:3 * h // 4, w // 3:2 * w // 3] = (255, 30, 200)
    img[::7, :, 1] = 255
    return img


def encode(img, **kw):
    b = io.BytesIO()
    Image.fromarray(img).save(b, format="JPEG", **kw)
    return b.getvalue()


def pillow_rgb(data):
    return np.array(Image.open(io.BytesIO(data)).convert("RGB"))


@pytest.mark.parametrize("size", [(16, 16), (8, 8), (17, 23), (48, 64), (90, 160), (1, 1), (15, 31), (21, 3), (5, 4),
                                  (33, 2), (2, 5), (3, 6)])
@pytest.mark.parametrize("sub", [0, 1, 2])
def test_oracle_decodes_like_libjpeg_turbo(size, sub):
    for kind in ("smooth", "noise", "edges"):
        for q in (50, 90, 100):
            data = encode(synth_image(size[0], size[1], kind), quality=q, subsampling=sub)
            assert np.array_equal(J.decode(data), pillow_rgb(data)), (size, sub, kind, q)


def test_oracle_grayscale_restart_and_optimised_tables():
    img = synth_image(40, 56, "noise", seed=3)
    img[:, :28] = synth_image(40, 28, "smooth")
    data = encode(img[..., 0], quality=85)
    assert J.parse(data)["comps"][0]["h"] == 1 and np.array_equal(J.decode(data), pillow_rgb(data))
    for sub in (0, 1, 2):
        for kw in (dict(restart_marker_blocks=1), dict(restart_marker_blocks=3), dict(restart_marker_rows=1),
                   dict(optimize=True)):
            data = encode(img, quality=80, subsampling=sub, **kw)
            if "optimize" not in kw:
                assert J.parse(data)["restart_interval"] > 0
            assert np.array_equal(J.decode(data), pillow_rgb(data)), (sub, kw)


def test_oracle_rejects_what_it_does_not_cover():
    img = synth_image(32, 32, "smooth")
    with pytest.raises(J.UnsupportedJpeg):
        J.parse(encode(img, progressive=True))
    with pytest.raises(J.UnsupportedJpeg):
        J.parse(b"\x00\x01")
    b = io.BytesIO()
    Image.fromarray(img).convert("CMYK").save(b, format="JPEG")
    with pytest.raises(J.UnsupportedJpeg):
        J.parse(b.getvalue())
