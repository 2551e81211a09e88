"""TEST INFRASTRUCTURE ONLY (see oracle/README in DESIGN.md section 5): CPU restatement of baseline JPEG decoding as
libjpeg-turbo does it with its default settings (JDCT_ISLOW, fancy upsampling, YCbCr -> RGB), which is what
PIL.Image.open(...).convert("RGB") and cv2.imread (reference: cama/reproject.py:224,243, cama/dataset_reader.py:74)
run.  Used by tests/ to check the device decoder stage by stage; never imported by the product.

The algorithm lives in a third-party library that is not part of /root/reference (opencv-python bundles libjpeg-turbo;
unpinned, requirements.txt:5).  It is restated here from the published algorithms -- ITU-T T.81 (Huffman / zigzag /
marker syntax), the IJG "islow" integer IDCT (Loeffler-Ligtenberg-Moschytz, 13-bit constants, 2 extra bits after
pass 1), IJG triangle ("fancy") chroma upsampling and the 16-bit fixed-point YCbCr->RGB tables -- and PINNED against
the real decoder: tests/test_oracle_jpeg.py compares every stage's end result byte-for-byte with Pillow's bundled
libjpeg-turbo on images of many sizes / subsamplings / qualities, here and on the GPU box.

Scope: baseline sequential DCT (SOF0), 8-bit, 1 or 3 components, one interleaved scan, luma sampling 1x1 / 2x1 / 2x2
with 1x1 chroma, restart intervals.  Anything else -> UnsupportedJpeg (callers fall back to the host decoder).
"""
import numpy as np


class UnsupportedJpeg(ValueError):
    pass


ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                   6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45,
                   38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int32)   # zigzag index -> natural index


# ------------------------------------------------------------------------------------------ container syntax (T.81 B)
def parse(data):
    """bytes -> dict(width, height, comps=[{id,h,v,tq,td,ta}], qt={id: (64,) natural order}, huff={(cls,id): (bits,
    vals)}, restart_interval, scan=bytes of the entropy-coded segment incl. stuffing and RSTn markers)."""
    data = bytes(data)
    if data[:2] != b"\xff\xd8":
        raise UnsupportedJpeg("no SOI")
    i, qt, huff, frame, dri = 2, {}, {}, None, 0
    while True:
        while data[i] != 0xFF:
            i += 1
        while data[i] == 0xFF:
            i += 1
        m = data[i]
        i += 1
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        if m == 0xD9:
            raise UnsupportedJpeg("EOI before SOS")
        L = (data[i] << 8) | data[i + 1]
        seg = data[i + 2:i + L]
        if m == 0xDB:
            j = 0
            while j < len(seg):
                pq, tq = seg[j] >> 4, seg[j] & 15
                j += 1
                if pq:
                    raise UnsupportedJpeg("16-bit quantisation table")
                t = np.zeros(64, np.int32)
                t[ZIGZAG] = np.frombuffer(seg[j:j + 64], np.uint8)
                qt[tq] = t
                j += 64
        elif m == 0xC4:
            j = 0
            while j < len(seg):
                tc, th = seg[j] >> 4, seg[j] & 15
                bits = np.frombuffer(seg[j + 1:j + 17], np.uint8).astype(np.int32)
                n = int(bits.sum())
                huff[(tc, th)] = (bits, np.frombuffer(seg[j + 17:j + 17 + n], np.uint8).astype(np.int32))
                j += 17 + n
        elif m == 0xC0:
            if seg[0] != 8:
                raise UnsupportedJpeg("precision")
            h, w, nc = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4], seg[5]
            comps = [{"id": seg[6 + 3 * k], "h": seg[7 + 3 * k] >> 4, "v": seg[7 + 3 * k] & 15, "tq": seg[8 + 3 * k]}
                     for k in range(nc)]
            frame = (w, h, comps)
        elif m in (0xC1, 0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise UnsupportedJpeg("not baseline sequential (SOF%d)" % (m - 0xC0))
        elif m == 0xDD:
            dri = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            if frame is None:
                raise UnsupportedJpeg("SOS before SOF")
            w, h, comps = frame
            ns = seg[0]
            if ns != len(comps):
                raise UnsupportedJpeg("non-interleaved scans")
            for k in range(ns):
                cid, t = seg[1 + 2 * k], seg[2 + 2 * k]
                if comps[k]["id"] != cid:
                    raise UnsupportedJpeg("scan component order")
                comps[k]["td"], comps[k]["ta"] = t >> 4, t & 15
            if tuple(seg[1 + 2 * ns:4 + 2 * ns]) != (0, 63, 0):
                raise UnsupportedJpeg("spectral selection / successive approximation")
            if len(comps) not in (1, 3):
                raise UnsupportedJpeg("component count")
            if len(comps) == 3 and (any((c["h"], c["v"]) != (1, 1) for c in comps[1:]) or
                                    (comps[0]["h"], comps[0]["v"]) not in ((1, 1), (2, 1), (2, 2))):
                raise UnsupportedJpeg("sampling factors")
            if len(comps) == 1:
                comps[0]["h"] = comps[0]["v"] = 1          # single-component scans are never interleaved (T.81 A.2.2)
            start = i + L
            # the entropy-coded segment ends at the first marker that is neither stuffing nor RSTn
            j = start
            while True:
                j = data.index(b"\xff", j)
                if data[j + 1] == 0 or 0xD0 <= data[j + 1] <= 0xD7:
                    j += 2
                    continue
                break
            return {"width": w, "height": h, "comps": comps, "qt": qt, "huff": huff, "restart_interval": dri,
                    "scan": data[start:j]}
        i += L


def geometry(hdr):
    """MCU grid: (hmax, vmax, mcus_x, mcus_y, blocks_per_mcu, per-component (blocks_w, blocks_h))."""
    hmax = max(c["h"] for c in hdr["comps"])
    vmax = max(c["v"] for c in hdr["comps"])
    mx = -(-hdr["width"] // (8 * hmax))
    my = -(-hdr["height"] // (8 * vmax))
    bpm = sum(c["h"] * c["v"] for c in hdr["comps"])
    return hmax, vmax, mx, my, bpm, [(mx * c["h"], my * c["v"]) for c in hdr["comps"]]


# ------------------------------------------------------------------------------------------ entropy decoding (T.81 F.2)
def huffman_lookup(bits, vals):
    """(maxcode[17], valptr[17], mincode[17]) per code length (T.81 F.2.2.3) + the symbol list."""
    code, k = 0, 0
    mincode, maxcode, valptr = np.zeros(17, np.int64), np.full(18, -1, np.int64), np.zeros(17, np.int64)
    for l in range(1, 17):
        if bits[l - 1]:
            valptr[l] = k
            mincode[l] = code
            code += int(bits[l - 1])
            k += int(bits[l - 1])
            maxcode[l] = code - 1
        code <<= 1
    return mincode, maxcode, valptr, vals


def unstuff(scan):
    """Entropy-coded bytes without 0xFF00 stuffing; RSTn markers removed, their byte positions (in the OUTPUT) returned."""
    a = np.frombuffer(scan, np.uint8)
    ff = np.flatnonzero(a[:-1] == 0xFF) if len(a) > 1 else np.zeros(0, np.int64)
    nxt = a[ff + 1]
    stuffed = ff[nxt == 0] + 1                       # the 0x00 bytes
    rst = ff[(nxt >= 0xD0) & (nxt <= 0xD7)]          # the 0xFF of each RSTn
    drop = np.zeros(len(a), bool)
    drop[stuffed] = True
    drop[rst] = True
    drop[rst + 1] = True
    keep_before = np.cumsum(~drop) - (~drop)
    return a[~drop].copy(), keep_before[rst].astype(np.int64)


def decode_coefficients(hdr):
    """Sequential Huffman decode -> int16 [n_blocks_total, 64] in SCAN order (MCU by MCU, component by component,
    natural coefficient order inside a block), DC prediction already undone."""
    hmax, vmax, mx, my, bpm, _ = geometry(hdr)
    data, rst_pos = unstuff(hdr["scan"])
    tabs = {k: huffman_lookup(*v) for k, v in hdr["huff"].items()}
    blk_comp = [ci for ci, c in enumerate(hdr["comps"]) for _ in range(c["h"] * c["v"])]
    nblocks = mx * my * bpm
    out = np.zeros((nblocks, 64), np.int16)
    nbits_total = len(data) * 8
    pos = 0                                           # bit position
    pred = [0] * len(hdr["comps"])
    ri = hdr["restart_interval"]
    rst_i = 0
    buf = int.from_bytes(bytes(data) + b"\0\0\0\0", "big")        # python big int: fine at test sizes
    total = nbits_total + 32

    def getbits(p, n):
        return (buf >> (total - p - n)) & ((1 << n) - 1) if n else 0

    def symbol(p, tab):
        mincode, maxcode, valptr, vals = tab
        code = 0
        for l in range(1, 17):
            code = (code << 1) | getbits(p + l - 1, 1)
            if maxcode[l] >= 0 and code <= maxcode[l] and code >= mincode[l]:
                return int(vals[valptr[l] + code - mincode[l]]), p + l
        raise ValueError("bad Huffman code at bit %d" % p)

    def extend(v, s):
        return v - ((1 << s) - 1) if s and v < (1 << (s - 1)) else v

    b = 0
    for mcu in range(mx * my):
        if ri and mcu and mcu % ri == 0:
            pos = int(rst_pos[rst_i]) * 8             # restart: byte-align at the marker, reset predictors
            rst_i += 1
            pred = [0] * len(hdr["comps"])
        for ci in blk_comp:
            c = hdr["comps"][ci]
            s, pos = symbol(pos, tabs[(0, c["td"])])
            diff = extend(getbits(pos, s), s)
            pos += s
            pred[ci] += diff
            out[b, 0] = pred[ci]
            k = 1
            ac = tabs[(1, c["ta"])]
            while k < 64:
                rs, pos = symbol(pos, ac)
                r, s = rs >> 4, rs & 15
                if s == 0:
                    if r != 15:
                        break
                    k += 16
                    continue
                k += r
                out[b, ZIGZAG[k]] = extend(getbits(pos, s), s)
                pos += s
                k += 1
            b += 1
    return out


# ------------------------------------------------------------------------------------------ IDCT (IJG jidctint, islow)
CONST_BITS, PASS1_BITS = 13, 2
F_0_298631336, F_0_390180644, F_0_541196100, F_0_765366865 = 2446, 3196, 4433, 6270
F_0_899976223, F_1_175875602, F_1_501321110, F_1_847759065 = 7373, 9633, 12299, 15137
F_1_961570560, F_2_053119869, F_2_562915447, F_3_072711026 = 16069, 16819, 20995, 25172


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(d, shift):
    """The LLM butterfly on the last axis of an int64 array [..., 8]; output descaled by `shift`."""
    z2, z3 = d[..., 2], d[..., 6]
    z1 = (z2 + z3) * F_0_541196100
    tmp2 = z1 + z3 * (-F_1_847759065)
    tmp3 = z1 + z2 * F_0_765366865
    z2, z3 = d[..., 0], d[..., 4]
    tmp0 = (z2 + z3) << CONST_BITS
    tmp1 = (z2 - z3) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = d[..., 7], d[..., 5], d[..., 3], d[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F_1_175875602
    tmp0 = tmp0 * F_0_298631336
    tmp1 = tmp1 * F_2_053119869
    tmp2 = tmp2 * F_3_072711026
    tmp3 = tmp3 * F_1_501321110
    z1 = z1 * (-F_0_899976223)
    z2 = z2 * (-F_2_562915447)
    z3 = z3 * (-F_1_961570560) + z5
    z4 = z4 * (-F_0_390180644) + z5
    tmp0 = tmp0 + z1 + z3
    tmp1 = tmp1 + z2 + z4
    tmp2 = tmp2 + z2 + z3
    tmp3 = tmp3 + z1 + z4
    return np.stack([_descale(tmp10 + tmp3, shift), _descale(tmp11 + tmp2, shift), _descale(tmp12 + tmp1, shift),
                     _descale(tmp13 + tmp0, shift), _descale(tmp13 - tmp0, shift), _descale(tmp12 - tmp1, shift),
                     _descale(tmp11 - tmp2, shift), _descale(tmp10 - tmp3, shift)], axis=-1)


def range_limit(v):
    """IJG range_limit[(v) & RANGE_MASK] with the +128 level shift: exact for |v| < 512 + wrap beyond, as the table."""
    x = np.asarray(v).astype(np.int64) & 1023
    return np.where(x < 128, x + 128, np.where(x < 512, 255, np.where(x < 896, 0, x - 896))).astype(np.uint8)


def idct_blocks(coef, q):
    """int16 [n,64] natural order, quant table (64,) -> uint8 [n,8,8] samples."""
    d = coef.astype(np.int64).reshape(-1, 8, 8) * q.astype(np.int64).reshape(8, 8)
    ws = _idct_1d(d.transpose(0, 2, 1), CONST_BITS - PASS1_BITS).transpose(0, 2, 1)        # pass 1: columns
    px = _idct_1d(ws, CONST_BITS + PASS1_BITS + 3)                                         # pass 2: rows
    return range_limit(px)


# ------------------------------------------------------------------------------------------ planes, upsampling, colour
def component_planes(hdr, coef):
    """Scan-order blocks -> per-component uint8 planes of the padded size (blocks_h*8, blocks_w*8)."""
    hmax, vmax, mx, my, bpm, dims = geometry(hdr)
    planes, off = [], 0
    for ci, c in enumerate(hdr["comps"]):
        n = c["h"] * c["v"]
        idx = (np.arange(mx * my)[:, None] * bpm + off + np.arange(n)[None, :])            # [mcu, blk in mcu]
        px = idct_blocks(coef[idx.reshape(-1)], hdr["qt"][c["tq"]]).reshape(my, mx, c["v"], c["h"], 8, 8)
        planes.append(px.transpose(0, 2, 4, 1, 3, 5).reshape(my * c["v"] * 8, mx * c["h"] * 8))
        off += n
    return planes


def upsample_h2v1(p, w_out):
    """IJG h2v1_fancy_upsample on a plane cropped to its downsampled size.  libjpeg-turbo selects the fancy routine
    only for downsampled widths > 2 (jdsample.c: `do_fancy && compptr->downsampled_width > 2`); narrower components
    are upsampled by plain replication."""
    if p.shape[1] <= 2:
        return np.repeat(p, 2, axis=1)[:, :w_out]
    a = p.astype(np.int32)
    left = np.concatenate([a[:, :1], a[:, :-1]], axis=1)
    right = np.concatenate([a[:, 1:], a[:, -1:]], axis=1)
    even = (a * 3 + left + 1) >> 2
    odd = (a * 3 + right + 2) >> 2
    even[:, 0] = a[:, 0]
    odd[:, -1] = a[:, -1]
    out = np.stack([even, odd], axis=-1).reshape(a.shape[0], -1)
    return out[:, :w_out].astype(np.uint8)


def upsample_h2v2(p, w_out, h_out):
    """IJG h2v2_fancy_upsample: 9/3/3/1 triangle filter, vertical neighbours replicated at the image edges; plain
    replication for downsampled widths <= 2 (see upsample_h2v1)."""
    if p.shape[1] <= 2:
        return np.repeat(np.repeat(p, 2, axis=0), 2, axis=1)[:h_out, :w_out]
    a = p.astype(np.int32)
    up = np.concatenate([a[:1], a[:-1]], axis=0)
    dn = np.concatenate([a[1:], a[-1:]], axis=0)
    rows = []
    for near in (up, dn):                                  # output row 2r uses the row above, 2r+1 the row below
        s = a * 3 + near                                   # column sums
        last = np.concatenate([s[:, :1], s[:, :-1]], axis=1)
        nxt = np.concatenate([s[:, 1:], s[:, -1:]], axis=1)
        even = (s * 3 + last + 8) >> 4
        odd = (s * 3 + nxt + 7) >> 4
        even[:, 0] = (s[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (s[:, -1] * 4 + 7) >> 4
        rows.append(np.stack([even, odd], axis=-1).reshape(a.shape[0], -1))
    out = np.stack(rows, axis=1).reshape(2 * a.shape[0], -1)
    return out[:h_out, :w_out].astype(np.uint8)


def ycc_to_rgb(y, cb, cr):
    """IJG jdcolor ycc_rgb_convert, 16-bit fixed point."""
    y = y.astype(np.int64)
    cbx, crx = cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = y + ((91881 * crx + 32768) >> 16)
    g = y + ((-22554 * cbx + 32768 - 46802 * crx) >> 16)
    b = y + ((116130 * cbx + 32768) >> 16)
    return np.stack([np.clip(r, 0, 255), np.clip(g, 0, 255), np.clip(b, 0, 255)], axis=-1).astype(np.uint8)


def decode_from_coefficients(hdr, coef):
    W, H = hdr["width"], hdr["height"]
    planes = component_planes(hdr, coef)
    if len(planes) == 1:
        g = planes[0][:H, :W]
        return np.stack([g, g, g], axis=-1)
    c0 = hdr["comps"][0]
    y = planes[0][:H, :W]
    if (c0["h"], c0["v"]) == (1, 1):
        cb, cr = planes[1][:H, :W], planes[2][:H, :W]
    elif (c0["h"], c0["v"]) == (2, 1):
        cw = -(-W // 2)
        cb, cr = (upsample_h2v1(p[:H, :cw], W) for p in planes[1:])
    else:
        cw, ch = -(-W // 2), -(-H // 2)
        cb, cr = (upsample_h2v2(p[:ch, :cw], W, H) for p in planes[1:])
    return ycc_to_rgb(y, cb, cr)


def decode(data):
    """JPEG bytes -> (H,W,3) uint8 RGB, what PIL.Image.open(...).convert("RGB") returns."""
    hdr = parse(data)
    return decode_from_coefficients(hdr, decode_coefficients(hdr))
