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
    for bad in (encode(img, progressive=True), b"not a jpeg"):
        with pytest.raises(PJ.Unsupported):
            PJ.parse_header(bad)
    # restart intervals: one entropy segment per interval, the markers themselves excluded
    data = encode(img, restart_marker_blocks=2, subsampling=2)
    h = PJ.parse_header(data)
    segs = PJ.restart_segments(data, h)
    assert h.restart_interval == 2 and len(segs) == 3 and [s[2:] for s in segs] == [(0, 2), (2, 2), (4, 2)]
    ref_scan = J.parse(data)["scan"]
    joined = b"".join(data[a:b] + (b"" if k == len(segs) - 1 else ref_scan[b - h.scan_start:b - h.scan_start + 2])
                      for k, (a, b, _, _) in enumerate(segs))
    assert joined == ref_scan and all(ref_scan[b - h.scan_start] == 0xFF for (_, b, _, _) in segs[:-1])


def test_descriptor_layout_and_plan():
    L = _lib.lib()
    assert PJ.IMAGE_DTYPE.itemsize == L.cama_jpeg_image_bytes() == 176
    imgs = np.zeros(2, PJ.IMAGE_DTYPE)
    for i, (w, h, hs, vs, ln) in enumerate([(1600, 900, 2, 2, 300000), (33, 17, 2, 1, 700)]):
        d = imgs[i]
        d["stream_off"], d["stream_len"] = (0 if i == 0 else 300080), ln
        d["width"], d["height"], d["ncomp"], d["hs"], d["vs"] = w, h, 3, hs, vs
        d["comp_dc"], d["comp_ac"] = [0, 1, 1], [0, 1, 1]
        d["out_slot"] = i
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
    imgs[1]["stream_off"] = 399999                           # runs past the stream bytes
    assert L.cama_jpeg_plan(imgs.ctypes.data, 2, 400000, info.ctypes.data) == -1


def test_plan_restart_interval_descriptors():
    """A pixels-only parent + one entropy segment per restart interval: segments write into the parent's coefficient
    block range and own no planes; the parent owns no workgroups."""
    L = _lib.lib()
    imgs = np.zeros(4, PJ.IMAGE_DTYPE)
    common = dict(ncomp=3, hs=2, vs=2, comp_dc=[0, 1, 1], comp_ac=[0, 1, 1])
    rows = [dict(common, kind=PJ.KIND_PIXELS, width=64, height=32, out_slot=0),
            dict(common, kind=PJ.KIND_SEGMENT, parent=0, first_block=0, width=3 * 16, height=16, stream_off=0, stream_len=500),
            dict(common, kind=PJ.KIND_SEGMENT, parent=0, first_block=18, width=3 * 16, height=16, stream_off=576, stream_len=300),
            dict(common, kind=PJ.KIND_SEGMENT, parent=0, first_block=36, width=2 * 16, height=16, stream_off=960, stream_len=900)]
    for r, row in enumerate(rows):
        for k, v in row.items():
            imgs[r][k] = v
    info = np.zeros(3, np.uint64)
    assert L.cama_jpeg_plan(imgs.ctypes.data, 4, 2048, info.ctypes.data) == 0, L.cama_last_error()
    assert imgs[0]["total_blocks"] == 4 * 2 * 6 and imgs[0]["nwg"] == 0 and imgs[0]["plane_w"][0] == 64
    assert [int(d["total_blocks"]) for d in imgs[1:]] == [18, 18, 12]
    assert [int(d["coef_off"]) for d in imgs[1:]] == [int(imgs[0]["coef_off"]) + 64 * b for b in (0, 18, 36)]
    assert all(d["plane_w"][0] == 0 for d in imgs[1:]) and [int(d["wg0"]) for d in imgs] == [0, 0, 1, 2]
    imgs[3]["first_block"] = 40                               # would run past the parent's blocks
    assert L.cama_jpeg_plan(imgs.ctypes.data, 4, 2048, info.ctypes.data) == -1 and b"parent" in L.cama_last_error()


def test_header_parser_hands_back_what_it_cannot_decode():
    """RGB-coded files (Adobe APP14 transform 0, or component ids 'R','G','B') and truncated headers raise Unsupported
    -- the caller then uses the host decoder -- instead of being decoded with the wrong colour transform or crashing."""
    import io
    from PIL import Image
    from cama_amd import jpeg as J
    img = Image.fromarray(np.random.default_rng(0).integers(0, 256, (16, 16, 3), dtype=np.uint8))
    b = io.BytesIO()
    img.save(b, "JPEG", quality=90)
    data = b.getvalue()
    assert J._parse_header(data).width == 16
    sof, sos = data.find(b"\xff\xc0"), data.find(b"\xff\xda")
    rgb = bytearray(data)
    for k, cid in enumerate(b"RGB"):
        rgb[sof + 10 + 3 * k] = cid
        rgb[sos + 5 + 2 * k] = cid
    adobe0 = data[:2] + b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x00" + data[2:]
    adobe1 = data[:2] + b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x01" + data[2:]
    assert J._parse_header(adobe1).width == 16                       # transform 1 = YCbCr: fine
    for bad in (bytes(rgb), adobe0, data[:sof + 8], data[:sos + 6], data[:data.find(b"\xff\xdb") + 30]):
        with pytest.raises(J.Unsupported):
            J._parse_header(bad)
