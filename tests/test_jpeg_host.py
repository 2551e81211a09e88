"""Host side of the device JPEG decoder (cama_amd/jpeg.py), no GPU: marker parsing against the oracle's own parser,
the Huffman table blob (10-bit lookup + per-length limits) against every code of the canonical tables, scope checks,
and the descriptor layout against the C ABI (sizeof + cama_jpeg_plan's derived fields)."""
import numpy as np
import pytest

from cama_amd import _lib
from cama_amd import jpeg as PJ
from oracle import jpeg_oracle as J
from tests.test_oracle_jpeg import encode, synth_image


def _lookup(rec, t, window32, second_level=False):
    """numpy mirror of jpeg_symbol() in jpeg_kernels.hpp: (length, symbol) for a 32-bit window; `second_level`: through the
    set's second-level table (jpeg_long_symbol) instead of the per-length limits."""
    e = int(rec["lut"][t, window32 >> 22])
    if e:
        return e >> 8, e & 255
    w16 = window32 >> 16
    if second_level:
        assert int(rec["l2_off"][t]) != PJ.L2_NONE
        e = int(rec["l2"][int(rec["l2_off"][t]) + min(w16, min(int(rec["lim"][t, 5]), 0xFFFF)) - (int(rec["l2_first"][t]) << 6)])
        return (16, None) if e == 16 << 8 else (e >> 8, e & 255)
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
                    assert _lookup(rec, t, (code << (32 - l)) | tail, True) == (l, int(vals[k])), (cls, tid, l, code)
                code += 1
                k += 1
            code <<= 1
        assert _lookup(rec, t, 0xFFFFFFFF) == (16, None)             # the all-ones prefix is never a code (T.81 C)
        assert _lookup(rec, t, 0xFFFFFFFF, True) == (16, None)
    # the second level covers exactly the 16-bit prefixes above the last 10-bit code, back to back, inside its budget
    at = 0
    for t in range(4):
        first = int(rec["l2_first"][t])
        assert (rec["lut"][t, :first] != 0).all() and (rec["lut"][t, first:] == 0).all()
        assert int(rec["l2_off"][t]) == at
        at += (min(int(rec["lim"][t, 5]), 0xFFFF) - (first << 6) + 1) if first < 1024 else 0
    assert at <= PJ.L2_MAX and (rec["l2"][:at] != 0).all() and (rec["l2"][at:] == 0).all()


def _sync_entry_scalar(is_ac, e):
    """T.81 F.2.2 state transition of one symbol entry (length << 8 | symbol): bits consumed | zigzag advance << 6."""
    ln, size, run = e >> 8, e & 15, (e >> 4) & 15
    kinc = ((run + 1) if size else (16 if run == 15 else 64)) if is_ac else 1
    return (ln + size) | (kinc << 6)


@pytest.mark.parametrize("kw", [dict(quality=90, subsampling=2), dict(quality=40, subsampling=0, optimize=True),
                                dict(quality=100, subsampling=1, optimize=True)])
def test_transition_tables_of_the_sync_phases_follow_the_symbol_tables(kw):
    """JpegSyncSet (built on the host since round 5, vectorised) against an entry-by-entry walk of its definition: a DC entry is
    its symbol's transition; an AC entry carries the symbol after it when that code lies inside the 10 known bits, the first
    does not end the block and both together consume at most 31 bits; the second level holds single transitions."""
    rec = PJ.build_huff_set(PJ.parse_header(encode(synth_image(67, 131, "noise", seed=4), **kw)).huff)
    n, two = 1 << PJ.LUT_BITS, 0
    for t in range(2):
        for x in range(n):
            e = int(rec["lut"][2 * t, x])
            assert int(rec["sync_dc"][t, x]) == (_sync_entry_scalar(False, e) if e else 0)
            e1 = int(rec["lut"][2 * t + 1, x])
            if not e1:
                assert int(rec["sync_ac"][t, x]) == 0
                continue
            s1 = _sync_entry_scalar(True, e1)
            s12, u1, k1 = s1, s1 & 63, s1 >> 6
            if k1 != 64 and u1 < PJ.LUT_BITS:
                e2 = int(rec["lut"][2 * t + 1, (x << u1) & (n - 1)])
                if e2 and (e2 >> 8) <= PJ.LUT_BITS - u1:
                    s2 = _sync_entry_scalar(True, e2)
                    if u1 + (s2 & 63) <= 31:
                        s12 = (u1 + (s2 & 63)) | ((k1 + (s2 >> 6)) << 6)
                        two += 1
            assert int(rec["sync_ac"][t, x]) == s1 | (s12 << 13), (t, x)
    assert two > 100                                                     # (the short codes do pair up)
    for t in range(4):
        if int(rec["l2_first"][t]) == n:
            continue
        lo = int(rec["l2_off"][t])
        hi = lo + min(int(rec["lim"][t, 5]), 0xFFFF) - (int(rec["l2_first"][t]) << 6) + 1
        for i in range(lo, hi):
            assert int(rec["sync_l2"][i]) == _sync_entry_scalar(t & 1, int(rec["l2"][i])), (t, i)
    assert (rec["sync_l2"][(rec["l2"] == 0)] == 0).all()


def test_table_set_whose_long_codes_do_not_fit_the_second_level_says_so():
    """A DHT with 200 codes of 16 bits below a short prefix: its second level would need > L2_MAX entries -> that table is
    left to the per-length limits (l2_off = L2_NONE), the others keep theirs."""
    bits = np.zeros(16, np.uint8)
    bits[1], bits[10], bits[15] = 2, 40, 120                          # 2 codes of 2 bits, 40 of 11, 120 of 16
    payload = bytes(bits) + bytes(range(162))
    std = PJ.build_huff_set({(0, 0): bytes([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]) + bytes(range(12)),
                             (1, 0): payload})
    assert int(std["l2_off"][0]) == 0 and int(std["l2_off"][1]) == PJ.L2_NONE
    assert int(std["lim"][1, 5]) - (int(std["l2_first"][1]) << 6) + 1 > PJ.L2_MAX
    # the tables after it still get theirs (DC1 / AC1 absent here: no codes at all -> the one "no code" entry each)
    n0 = int(std["lim"][0, 5]) - (int(std["l2_first"][0]) << 6) + 1
    assert [int(v) for v in std["l2_off"][2:]] == [n0, n0 + 1]


def _kernel_l2_adjust(rec, t):
    """Python twin of jpeg_l2_adjust1 (jpeg_kernels.hpp): top << 16 | adj, 0 for a table whose second level is never used,
    None = JPEG_L2_ABSENT (slow path)."""
    lim5, first, off = int(rec["lim"][t, 5]), int(rec["l2_first"][t]), int(rec["l2_off"][t])
    top, base = min(lim5, 0xFFFF), first << 6
    if lim5 == 0 or first >= (1 << PJ.LUT_BITS):
        return 0
    if off == PJ.L2_NONE or base < off or base - off > 0xFFFF:
        return None
    return (top << 16) | (base - off)


def _kernel_l2_index(adj, w16):
    return max(min(w16, adj >> 16) - (adj & 0xFFFF), 0)


def test_second_level_index_stays_inside_the_table_for_every_table_of_any_set():
    """a grayscale file (or one that defines only the DC0 / AC0 pair) leaves tables 2 and 3 without codes; the
    kernels' wave-mode walk evaluates all four tables on every lane, so the rebased index must be in bounds for ANY 16-bit
    prefix under ANY table -- and agree with the per-length limits wherever a long code exists."""
    gray = encode(synth_image(40, 56, "noise", seed=2).mean(axis=2).astype(np.uint8), quality=85)
    colour = encode(synth_image(40, 56, "noise", seed=3), quality=92, subsampling=2, optimize=True)
    bits = np.zeros(16, np.uint8)
    bits[15] = 3                                                      # a table of three 16-bit codes and nothing else
    only_long = {(0, 0): bytes(bits) + bytes([1, 2, 3]), (1, 0): bytes(bits) + bytes([0x11, 0x12, 0x00])}
    full = np.zeros(16, np.uint8)
    full[0:15] = 1
    full[15] = 2                                                      # Kraft sum 1: the prefix 0xFFFF is a code
    sets = [PJ.build_huff_set(PJ.parse_header(gray).huff), PJ.build_huff_set(PJ.parse_header(colour).huff),
            PJ.build_huff_set(only_long), PJ.build_huff_set({(0, 0): bytes(full) + bytes(range(17))})]
    probes = np.unique(np.concatenate([np.arange(0, 0x10000, 97), np.arange(0xFF00, 0x10000), [0, 1, 63, 64]]))
    seen_absent = seen_unused = 0
    for rec in sets:
        for t in range(4):
            adj = _kernel_l2_adjust(rec, t)
            if adj is None:
                seen_absent += 1
                continue
            seen_unused += adj == 0
            for w16 in probes:
                i = _kernel_l2_index(adj, int(w16))
                assert 0 <= i < PJ.L2_MAX, (t, w16, i)
                if adj and int(rec["lut"][t, int(w16) >> 6]) == 0:      # a long code (or none) starts here: same answer as the limits
                    assert int(rec["l2"][i]) == int(PJ.long_symbol(rec, t, [int(w16)])[0]), (t, hex(int(w16)))
    assert seen_unused >= 2 and seen_absent >= 1                      # grayscale's tables 2/3; the long-codes-only table
    # the fully subscribed table keeps its last code: no "no code" sentinel over prefix 0xFFFF
    rec = sets[3]
    assert int(rec["lim"][0, 5]) == 0x10000
    assert int(rec["l2"][_kernel_l2_index(_kernel_l2_adjust(rec, 0), 0xFFFF)]) == int(PJ.long_symbol(rec, 0, [0xFFFF])[0]) != 16 << 8


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


def _marker_positions(buf):
    """What cama_jpeg_find_restarts returns (sorted): every 0xFF 0xD0..0xD7 pair."""
    idx = np.flatnonzero(buf[:-1] == 0xFF)
    return idx[(buf[idx + 1] & 0xF8) == 0xD0].astype(np.int64)


def test_batched_restart_segmentation_equals_the_per_file_walk():
    """segments_from_markers (the group-at-once route fed by the device's marker search) == restart_segments (the
    per-file walk) on whole files laid end to end -- headers in between included, as in an arena upload -- and drops
    exactly the scans whose markers are missing, out of cycle, or adjacent (empty interval)."""
    rng = np.random.default_rng(11)
    files = []
    for (h, w) in [(64, 96), (203, 317), (33, 47), (120, 200)]:
        img = synth_image(h, w, "noise", seed=h)
        for sub in (0, 1, 2):
            for kw in (dict(restart_marker_blocks=1), dict(restart_marker_blocks=7), dict(restart_marker_rows=1)):
                files.append(encode(img, quality=int(rng.integers(30, 100)), subsampling=sub, **kw))
    files.append(encode(synth_image(40, 40, "smooth"), restart_marker_blocks=10000))       # one interval, no marker
    corrupt = {}
    for kind, j in (("missing", 3), ("cycle", 8), ("empty", 14), ("extra", 20)):
        data, hd = bytearray(files[j]), PJ.parse_header(files[j])
        segs = PJ.restart_segments(bytes(data), hd)
        m = segs[1][1]                                                # second marker
        if kind == "missing":
            data[m:m + 2] = b"\x12\x34"
        elif kind == "cycle":
            data[m + 1] = 0xD0 | ((data[m + 1] + 3) & 7)
        elif kind == "empty":
            data[m + 2:m + 2] = bytes((0xFF, 0xD0 | ((data[m + 1] + 1) & 7)))        # two markers back to back
        else:
            data[segs[-1][0] + 1:segs[-1][0] + 1] = b"\xff\xd3"                        # one marker too many
        files[j] = bytes(data)
        corrupt[j] = kind
    headers = [PJ.parse_header(f) for f in files]
    offs = np.concatenate([[0], np.cumsum([len(f) + 16 for f in files])[:-1]])
    buf = np.zeros(int(offs[-1]) + len(files[-1]) + 16, np.uint8)
    for o, f in zip(offs, files):
        buf[o:o + len(f)] = np.frombuffer(f, np.uint8)
    lo = np.array([o + h.scan_start for o, h in zip(offs, headers)], dtype=np.int64)
    hi = np.array([o + h.scan_end for o, h in zip(offs, headers)], dtype=np.int64)
    mcus = np.array([-(-h.width // (8 * h.hs)) * -(-h.height // (8 * h.vs)) for h in headers])
    nseg = -(-mcus // np.array([h.restart_interval for h in headers]))
    good = np.ones(len(files), bool)
    g, ns, k, starts, ends = PJ.segments_from_markers(_marker_positions(buf), buf, lo, hi, nseg, good)
    assert sorted(np.flatnonzero(~good).tolist()) == sorted(corrupt), (np.flatnonzero(~good), corrupt)
    at = 0
    for j, n in zip(g.tolist(), ns.tolist()):
        want = PJ.restart_segments(files[j], headers[j])
        assert want is not None and len(want) == n
        assert [(int(a) - offs[j], int(b) - offs[j]) for a, b in zip(starts[at:at + n], ends[at:at + n])] == \
            [(a, b) for a, b, _, _ in want], j
        assert k[at:at + n].tolist() == list(range(n))
        at += n
    assert at == len(k)
    for j in corrupt:                                                 # the per-file walk refuses them too
        assert PJ.restart_segments(files[j], headers[j]) is None or corrupt[j] == "cycle", corrupt[j]


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
