"""Host side of the device JPEG decoder (cama_amd/jpeg.py), no GPU: marker parsing against the oracle's own parser,
the Huffman table blob (10-bit lookup + per-length limits) against every code of the canonical tables, scope checks,
and the descriptor layout against the C ABI (sizeof + cama_jpeg_plan's derived fields)."""
import numpy as np
import pytest

from cama_amd import _lib
from cama_amd import jpeg as PJ
from oracle import jpeg_oracle as J
from tests.test_oracle_jpeg import encode, synth_image


def _lookup(rec, t, window32):
    """numpy mirror of jpeg_symbol() in jpeg_kernels.hpp: (length, symbol) for a 32-bit window."""
    e = int(rec["lut"][t, window32 >> 22])
    if e:
        return e >> 8, e & 255
    w16 = window32 >> 16
    lim = rec["lim"][t]
    if w16 >= lim[5]:
        return 16, None
    l = 11 + sum(int(w16 >= lim[i]) for i in range(5))
    return l, int(rec["vals"][t, ((w16 >> (16 - l)) + int(rec["valoff"][t, l])) & 255])


@pytest.mark.parametrize("kw", [dict(quality=90, subsampling=2), dict(quality=40, subsampling=0, optimize=True),
                                dict(quality=100, subsampling=1, optimize=True)])
def test_header_and_tables_agree_with_the_oracle(kw):
    data = encode(synth_image(67, 131, "noise", seed=4), **kw)
    PJ._HEADER_CACHE.clear()
    h, ref = PJ.parse_header(data), J.parse(data)
    again = PJ.parse_header(data)                                   # second call: served from the per-camera cache
    assert (again.scan_start, again.scan_end, again.width) == (h.scan_start, h.scan_end, h.width) and len(PJ._HEADER_CACHE) == 1
    assert (h.width, h.height, h.ncomp) == (ref["width"], ref["height"], len(ref["comps"]))
    assert (h.hs, h.vs) == (ref["comps"][0]["h"], ref["comps"][0]["v"])
    assert data[h.scan_start:h.scan_end] == ref["scan"]
    for k, c in enumerate(ref["comps"]):
        assert np.array_equal(h.quant[k], ref["qt"][c["tq"]]) and (h.comp_dc[k], h.comp_ac[k]) == (c["td"], c["ta"])
    rec = PJ.build_huff_set(h.huff)
    assert rec.dtype.itemsize == _lib.lib().cama_jpeg_huff_set_bytes()
    for (cls, tid), (bits, vals) in ref["huff"].items():
        t = tid * 2 + cls
        code = k = 0
        for l in range(1, 17):
            for _ in range(int(bits[l - 1])):
                for tail in (0, (1 << (32 - l)) - 1):                # the code followed by all zeros / all ones
                    assert _lookup(rec, t, (code << (32 - l)) | tail) == (l, int(vals[k])), (cls, tid, l, code)
                code += 1
                k += 1
            code <<= 1
        assert _lookup(rec, t, 0xFFFFFFFF) == (16, None)             # the all-ones prefix is never a code (T.81 C)


def test_scope_checks():
    img = synth_image(32, 48, "smooth")
    for bad in (encode(img, progressive=True), encode(img, restart_marker_blocks=1), b"not a jpeg"):
        with pytest.raises(PJ.Unsupported):
            PJ.parse_header(bad)


def test_descriptor_layout_and_plan():
    L = _lib.lib()
    assert PJ.IMAGE_DTYPE.itemsize == L.cama_jpeg_image_bytes() == 160
    imgs = np.zeros(2, PJ.IMAGE_DTYPE)
    for i, (w, h, hs, vs, ln) in enumerate([(1600, 900, 2, 2, 300000), (33, 17, 2, 1, 700)]):
        d = imgs[i]
        d["stream_off"], d["stream_len"] = (0 if i == 0 else 300080), ln
        d["width"], d["height"], d["ncomp"], d["hs"], d["vs"] = w, h, 3, hs, vs
        d["comp_dc"], d["comp_ac"] = [0, 1, 1], [0, 1, 1]
    info = np.zeros(3, np.uint64)
    assert L.cama_jpeg_plan(imgs.ctypes.data, 2, 300080 + 700 + 64, info.ctypes.data) == 0, L.cama_last_error()
    a, b = imgs[0], imgs[1]
    assert (a["mx"], a["my"], a["bpm"], a["total_blocks"]) == (100, 57, 6, 100 * 57 * 6)
    assert (b["mx"], b["my"], b["bpm"], b["total_blocks"]) == (3, 3, 4, 36)
    assert a["nwg"] == -(-(300000 * 8) // (1024 * 256)) and b["wg0"] == a["nwg"] and b["tile0"] == a["ntile"] == 293
    assert tuple(a["plane_w"]) == (1600, 800, 800) and tuple(a["plane_h"]) == (912, 456, 456)
    assert b["coef_off"] == a["total_blocks"] * 64 and int(info[0]) > 0
    # bad input: unsupported sampling, overlapping segments
    imgs[1]["vs"] = 4
    assert L.cama_jpeg_plan(imgs.ctypes.data, 2, 400000, info.ctypes.data) == -1 and b"sampling" in L.cama_last_error()
    imgs[1]["vs"] = 1
    imgs[1]["stream_off"] = 16
    assert L.cama_jpeg_plan(imgs.ctypes.data, 2, 400000, info.ctypes.data) == -1
