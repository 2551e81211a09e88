"""Device JPEG decode for frame ingest: the host side of cama_jpeg_decode (include/cama_hip.h).

The reference decodes every camera frame with cv2.imread on one host core (cama/reproject.py:224,243;
cama/dataset_reader.py:72-76) -- ~10 ms per 1600x900 image, 96 % of a demo frame once the reprojection runs on the GPU.
Here the host only walks the marker segments (a few dozen bytes of headers per file), packs the entropy-coded bytes of
a whole batch into one upload and hands descriptors + tables to the device decoder, whose output is byte-identical to
libjpeg-turbo's (tests/test_gpu_jpeg.py against Pillow and oracle/jpeg_oracle.py).

Scope of the device path: baseline sequential (SOF0), 8 bit, grey or YCbCr with 4:4:4 / 4:2:2 / 4:2:0 sampling, one
interleaved scan, with or without restart intervals (every interval becomes its own entropy segment).  Anything else -- and any stream the device flags as inconsistent -- is decoded
by Pillow on the host and uploaded, so `decode()` always returns every image.
"""
import io
import os

import numpy as np

from . import _lib

IMAGE_DTYPE = np.dtype([
    ("stream_off", "<u8"), ("clean_off", "<u8"), ("coef_off", "<u8"), ("plane_off", "<u8", (3,)),
    ("stream_len", "<u4"), ("width", "<u4"), ("height", "<u4"), ("ncomp", "<u4"), ("hs", "<u4"), ("vs", "<u4"),
    ("huff_set", "<u4"), ("quant_set", "<u4"), ("comp_dc", "<u4", (3,)), ("comp_ac", "<u4", (3,)),
    ("mx", "<u4"), ("my", "<u4"), ("bpm", "<u4"), ("total_blocks", "<u4"),
    ("wg0", "<u4"), ("nwg", "<u4"), ("tile0", "<u4"), ("ntile", "<u4"),
    ("plane_w", "<u4", (3,)), ("plane_h", "<u4", (3,)),
    ("kind", "<u4"), ("parent", "<u4"), ("first_block", "<u4"), ("out_slot", "<u4")])       # == cama_jpeg_image
KIND_WHOLE, KIND_SEGMENT, KIND_PIXELS = 0, 1, 2

LUT_BITS = 10
L2_MAX, L2_NONE = 1024, 0xFFFF
HUFF_DTYPE = np.dtype([("lut", "<u2", (4, 1 << LUT_BITS)), ("lim", "<u4", (4, 8)), ("valoff", "<i4", (4, 17)),
                       ("vals", "u1", (4, 256)), ("l2_first", "<u2", (4,)), ("l2_off", "<u2", (4,)),
                       ("l2", "<u2", (L2_MAX,)),                                              # == JpegHuffSet, then
                       ("sync_dc", "<u2", (2, 1 << LUT_BITS)), ("sync_ac", "<u4", (2, 1 << LUT_BITS)),
                       ("sync_l2", "<u2", (L2_MAX,))])                                        # JpegSyncSet (== JpegHuffRec)

_ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
                    13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52,
                    45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])


class Unsupported(ValueError):
    """The file is outside the device decoder's scope; use the host decoder."""


class JpegHeader:
    __slots__ = ("width", "height", "ncomp", "hs", "vs", "quant", "huff", "comp_dc", "comp_ac", "scan_start",
                 "scan_end", "restart_interval")


_HEADER_CACHE = {}


class PinnedArena:
    """A pinned host buffer that compressed files are read INTO (file -> pinned memory, one copy), so that a batch's
    entropy data goes to the device with one asynchronous copy of a contiguous span instead of being packed into a
    staging buffer first (a Python-side memcpy of ~75 MB per 240 photo-like frames: it was 90 % of a batch's host time
    and made the decoder host-bound at 31 k images/s).  Bump allocation, 16-byte aligned slices."""

    def __init__(self, nbytes, pool=None):
        import torch
        self.buf = torch.empty(int(nbytes), dtype=torch.uint8).pin_memory()
        self.np = self.buf.numpy()
        self.size, self.used, self._pool = int(nbytes), 0, pool

    def take(self, n):
        """An ArenaBlob of n bytes, or None when the arena is full."""
        off = (self.used + 15) & ~15
        if off + n > self.size:
            return None
        self.used = off + n
        return ArenaBlob(self, off, n)


class ArenaBlob:
    """n bytes of a PinnedArena holding one compressed file (filled by the reader)."""
    __slots__ = ("arena", "off", "n")

    def __init__(self, arena, off, n):
        self.arena, self.off, self.n = arena, off, n

    def view(self):
        return self.arena.np[self.off:self.off + self.n]

    def tobytes(self):
        return self.view().tobytes()

    def __len__(self):
        return self.n


def is_blob(x):
    return isinstance(x, (bytes, bytearray, ArenaBlob))


def as_bytes(x):
    return x.tobytes() if isinstance(x, ArenaBlob) else x


def parse_header_blob(blob):
    """parse_header for an ArenaBlob without copying the file: the header is looked up on its first bytes, the scan's end
    on its last bytes (EOI is the file's tail); only unusual layouts fall back to a full copy."""
    v = blob.view()
    n = len(v)
    head = v[:min(n, 8192)].tobytes()
    sos = head.find(b"\xff\xda")
    if sos > 0 and sos + 4 <= len(head):
        key = head[:sos + 2 + ((head[sos + 2] << 8) | head[sos + 3])]
        hit = _HEADER_CACHE.get(key)
        if hit is not None and len(key) <= len(head):
            tail0 = max(0, n - 64)
            e = v[tail0:].tobytes().rfind(b"\xff\xd9")
            if e >= 0 and tail0 + e >= hit.scan_start:
                h = JpegHeader()
                for name in JpegHeader.__slots__:
                    setattr(h, name, getattr(hit, name))
                h.scan_end = tail0 + e
                return h
    return parse_header(v.tobytes())


def parse_header(data):
    """Header of a JPEG file; the frames of one camera share every byte up to the scan, so the parsed header is cached
    on those bytes and only the scan's end is looked up per file."""
    sos = data.find(b"\xff\xda")
    if sos > 0 and sos + 4 <= len(data):
        key = bytes(data[:sos + 2 + ((data[sos + 2] << 8) | data[sos + 3])])
        hit = _HEADER_CACHE.get(key)
        if hit is None:
            hit = _parse_header(data)
            if len(_HEADER_CACHE) < 256 and hit.scan_start == len(key):
                _HEADER_CACHE[key] = hit
            return hit
        h = JpegHeader()
        for name in JpegHeader.__slots__:
            setattr(h, name, getattr(hit, name))
        h.scan_end = data.rfind(b"\xff\xd9")
        if h.scan_end < h.scan_start:
            raise Unsupported("no EOI")
        return h
    return _parse_header(data)


def _parse_header(data):
    """Walk the marker segments up to SOS (T.81 B.2); anything this decoder does not handle -- including truncated or
    malformed segments -- raises Unsupported, which sends the image to the host decoder."""
    try:
        return _parse_header_strict(data)
    except Unsupported:
        raise
    except (IndexError, ValueError, KeyError) as e:          # truncated DQT / DHT / SOF / SOS payloads
        raise Unsupported(f"malformed header: {e}") from None


def _parse_header_strict(data):
    """Returns a JpegHeader whose `quant` is (3,64) uint16 in natural order, `huff` the raw DHT payloads
    {(class, id): bytes}, and [scan_start, scan_end) the entropy-coded bytes."""
    if data[:2] != b"\xff\xd8":
        raise Unsupported("not a JPEG")
    n = len(data)
    i = 2
    qt, huff, frame, dri = {}, {}, None, 0
    adobe_transform = None
    while i + 4 <= n:
        if data[i] != 0xFF:
            raise Unsupported("marker expected")
        m = data[i + 1]
        if m == 0xFF:
            i += 1
            continue
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            i += 2
            continue
        L = (data[i + 2] << 8) | data[i + 3]
        seg = data[i + 4:i + 2 + L]
        if m == 0xDB:
            j = 0
            while j < len(seg):
                if seg[j] >> 4:
                    raise Unsupported("16-bit quantisation table")
                qt[seg[j] & 15] = np.frombuffer(seg, np.uint8, 64, j + 1)
                j += 65
        elif m == 0xC4:
            j = 0
            while j < len(seg):
                cnt = sum(seg[j + 1:j + 17])
                huff[(seg[j] >> 4, seg[j] & 15)] = bytes(seg[j + 1:j + 17 + cnt])
                j += 17 + cnt
        elif m == 0xC0:
            frame = seg
        elif 0xC1 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise Unsupported("not baseline sequential DCT")
        elif m == 0xDD:
            dri = (seg[0] << 8) | seg[1]
        elif m == 0xEE and len(seg) >= 12 and bytes(seg[:5]) == b"Adobe":
            adobe_transform = seg[11]                        # APP14: 0 = no colour transform (RGB / CMYK), 1 = YCbCr
        elif m == 0xDA:
            if frame is None:
                raise Unsupported("SOS before SOF")
            break
        elif m == 0xD9:
            raise Unsupported("no scan")
        i += 2 + L
    else:
        raise Unsupported("truncated")
    h = JpegHeader()
    if frame[0] != 8:
        raise Unsupported("sample precision")
    h.height, h.width, h.ncomp = (frame[1] << 8) | frame[2], (frame[3] << 8) | frame[4], frame[5]
    if h.ncomp not in (1, 3) or h.width == 0 or h.height == 0:
        raise Unsupported("component count / size")
    comps = [(frame[6 + 3 * k], frame[7 + 3 * k] >> 4, frame[7 + 3 * k] & 15, frame[8 + 3 * k]) for k in range(h.ncomp)]
    h.restart_interval = dri
    if h.ncomp == 3:
        # libjpeg's colour-space guess (jdapimin.c default_decompress_parms): Adobe transform 0 or component ids
        # 'R','G','B' mean the samples are RGB already -- this decoder always applies YCbCr -> RGB, so hand those back
        ids = tuple(c[0] for c in comps)
        if adobe_transform == 0 or (adobe_transform is None and ids == (0x52, 0x47, 0x42)):
            raise Unsupported("RGB-coded JPEG (no YCbCr transform)")
        if adobe_transform not in (None, 0, 1):
            raise Unsupported("unknown Adobe colour transform")
    if seg[0] != h.ncomp or tuple(seg[1 + 2 * h.ncomp:4 + 2 * h.ncomp]) != (0, 63, 0):
        raise Unsupported("not one interleaved full-spectrum scan")
    h.comp_dc, h.comp_ac = [0, 0, 0], [0, 0, 0]
    for k in range(h.ncomp):
        if seg[1 + 2 * k] != comps[k][0]:
            raise Unsupported("scan component order")
        h.comp_dc[k], h.comp_ac[k] = seg[2 + 2 * k] >> 4, seg[2 + 2 * k] & 15
        if h.comp_dc[k] > 1 or h.comp_ac[k] > 1 or (0, h.comp_dc[k]) not in huff or (1, h.comp_ac[k]) not in huff:
            raise Unsupported("Huffman table selector")
    if h.ncomp == 1:
        h.hs = h.vs = 1
    else:
        h.hs, h.vs = comps[0][1], comps[0][2]
        if (h.hs, h.vs) not in ((1, 1), (2, 1), (2, 2)) or any(c[1:3] != (1, 1) for c in comps[1:]):
            raise Unsupported("sampling factors")
    h.quant = np.zeros((3, 64), np.uint16)
    for k in range(h.ncomp):
        if comps[k][3] not in qt:
            raise Unsupported("missing quantisation table")
        h.quant[k, _ZIGZAG] = qt[comps[k][3]]
    h.huff = huff
    h.scan_start = i + 2 + L
    end = data.rfind(b"\xff\xd9")
    if end < h.scan_start:
        raise Unsupported("no EOI")
    h.scan_end = end
    return h


def build_huff_set(huff):
    """{(class, id): DHT payload (16 counts + symbols)} -> one HUFF_DTYPE record: tables DC0, AC0, DC1, AC1 with a
    10-bit lookup (length << 8 | symbol) and, for longer codes, per-length 16-bit limits + value offsets (the canonical
    code structure of T.81 C / F.2.2.3: a code's length is 11 + the number of limits its 16-bit prefix has reached)."""
    rec = np.zeros((), HUFF_DTYPE)
    for (cls, tid), payload in huff.items():
        if tid > 1:
            continue
        t = tid * 2 + cls
        bits = np.frombuffer(payload, np.uint8, 16)
        vals = np.frombuffer(payload, np.uint8, offset=16)
        rec["vals"][t, :len(vals)] = vals
        code = k = 0
        for l in range(1, 17):
            cnt = int(bits[l - 1])
            if cnt:
                rec["valoff"][t, l] = k - code
                if l <= LUT_BITS:
                    span = 1 << (LUT_BITS - l)
                    entries = (np.uint16(l) << 8) | vals[k:k + cnt].astype(np.uint16)
                    rec["lut"][t, code << (LUT_BITS - l):(code + cnt) << (LUT_BITS - l)] = np.repeat(entries, span)
                code += cnt
                k += cnt
            if l > LUT_BITS:
                # exclusive upper limit of all codes of length <= l, left-aligned to 16 bits (monotone in l)
                rec["lim"][t, l - LUT_BITS - 1] = code << (16 - l)
            code <<= 1
    _second_level(rec)
    _sync_tables(rec)
    return rec


def _sync_entry(is_ac, e):
    """Transition entry of a symbol entry e = (length << 8) | symbol, e != 0 (array): bits consumed | zigzag advance << 6 -- DC: 1;
    AC with a value: run + 1; ZRL: 16; EOB: 64 (T.81 F.2.2; jpeg_kernels.hpp: jpeg_sync_entry)."""
    e = np.asarray(e, np.int64)
    ln, size, run = e >> 8, e & 15, (e >> 4) & 15
    kinc = np.where(size != 0, run + 1, np.where(run == 15, 16, 64)) if is_ac else np.ones_like(e)
    return (ln + size) | (kinc << 6)


def _sync_tables(rec):
    """The synchronisation phases' tables (jpeg_kernels.hpp: JpegSyncSet, jpeg_sync_span): they only need a symbol's bit count
    and where it leaves the zigzag index.  An AC entry also carries the symbol AFTER it when that one's code lies inside the
    10 known bits too and the first does not end the block: used1 | kinc1 << 6 | used12 << 13 | kinc12 << 19 (used12 = used1,
    kinc12 = kinc1: no second symbol).  They depend on the table set alone, so they are built here, once per set, and every
    workgroup copies them (until round 5 every workgroup derived them on the device)."""
    n = 1 << LUT_BITS
    x = np.arange(n, dtype=np.int64)
    for t in range(2):
        e = rec["lut"][2 * t].astype(np.int64)
        rec["sync_dc"][t] = np.where(e != 0, _sync_entry(False, np.maximum(e, 1)), 0)
        lut = rec["lut"][2 * t + 1].astype(np.int64)
        e1 = lut
        s1 = _sync_entry(True, np.maximum(e1, 1))
        u1, k1 = s1 & 63, s1 >> 6
        first = (e1 != 0) & (k1 != 64) & (u1 < LUT_BITS)
        e2 = lut[(x << np.where(first, u1, 0)) & (n - 1)]                  # the bits behind symbol 1, zero-padded
        second = first & (e2 != 0) & ((e2 >> 8) <= LUT_BITS - u1)           # its code lies inside the known bits
        s2 = _sync_entry(True, np.maximum(e2, 1))
        u12, k12 = u1 + (s2 & 63), k1 + (s2 >> 6)
        second &= u12 <= 31                                                # (the bit window advances <= 31 bits a step)
        s12 = np.where(second, u12 | (k12 << 6), s1)
        rec["sync_ac"][t] = np.where(e1 != 0, s1 | (s12 << 13), 0)
    # second level: table t's entries are l2[l2_off[t], +its span); AC tables are 1 and 3
    l2 = rec["l2"].astype(np.int64)
    is_ac = np.zeros(L2_MAX, bool)
    for t in (1, 3):
        if int(rec["l2_off"][t]) != L2_NONE and int(rec["l2_first"][t]) != n:
            lo = int(rec["l2_off"][t])
            hi = lo + min(int(rec["lim"][t, 5]), 0xFFFF) - (int(rec["l2_first"][t]) << 6) + 1
            is_ac[lo:hi] = True
    safe = np.maximum(l2, 1)
    rec["sync_l2"] = np.where(l2 != 0, np.where(is_ac, _sync_entry(True, safe), _sync_entry(False, safe)), 0)


def long_symbol(rec, t, w16):
    """(length << 8) | symbol of the code of 11..16 bits that starts the 16-bit prefixes `w16` (array) under table t; 16 << 8
    where no code starts there -- jpeg_kernels.hpp: jpeg_symbol_long_t, vectorised."""
    w16 = np.asarray(w16, np.int64)
    lim = rec["lim"][t].astype(np.int64)
    l = 11 + sum((w16 >= lim[i]).astype(np.int64) for i in range(5))
    idx = ((w16 >> (16 - l)) + rec["valoff"][t].astype(np.int64)[l]) & 255
    sym = (l << 8) | rec["vals"][t].astype(np.int64)[idx]
    return np.where(w16 >= lim[5], 16 << 8, sym).astype(np.uint16)


def _second_level(rec):
    """Second-level lookup of the codes longer than LUT_BITS: canonical codes grow with their length, so they start at the
    first 10-bit prefix whose lut entry is 0 (l2_first) and end where the code space of the table ends (lim[5], the exclusive
    limit of all its codes as a 16-bit prefix); one entry per 16-bit prefix in between plus ONE for everything above (no
    code: 16 << 8) -- the kernels clamp the prefix to that last entry.  A set whose long codes do not fit L2_MAX entries keeps
    l2_off = L2_NONE for the tables left out (the kernels then walk the per-length limits instead)."""
    at = 0
    for t in range(4):
        zero = np.flatnonzero(rec["lut"][t] == 0)
        first = int(zero[0]) if len(zero) else 1 << LUT_BITS
        assert not len(zero) or len(zero) == (1 << LUT_BITS) - first, "long codes are a suffix of the prefix space"
        rec["l2_first"][t] = first
        if not len(zero):
            rec["l2_off"][t] = at                                    # (never looked up)
            continue
        base, top = first << (16 - LUT_BITS), min(int(rec["lim"][t, 5]), 0xFFFF)
        n = top - base + 1
        assert n >= 1
        if at + n > L2_MAX:
            rec["l2_off"][t] = L2_NONE
            continue
        rec["l2_off"][t] = at
        rec["l2"][at:at + n] = long_symbol(rec, t, base + np.arange(n))
        if int(rec["lim"][t, 5]) <= 0xFFFF:                          # (a fully subscribed table has a code at prefix 0xFFFF too)
            rec["l2"][at + n - 1] = 16 << 8                          # the clamp target: no code from here on
        at += n


_RST = [bytes((0xFF, 0xD0 + k)) for k in range(8)]


def restart_segments(data, h):
    """[(start, end, first_mcu, n_mcu)] byte ranges of the restart intervals of the scan (T.81 B.2.1: the markers
    RST0..RST7 cycle between them; 0xFF 0xDk cannot occur inside entropy-coded data), or None when their number does
    not match the frame (corrupt: leave it to the host decoder)."""
    mcus = -(-h.width // (8 * h.hs)) * -(-h.height // (8 * h.vs))
    ri = h.restart_interval
    nseg = -(-mcus // ri)
    pos, segs = h.scan_start, []
    for k in range(nseg - 1):
        m = data.find(_RST[k & 7], pos, h.scan_end)
        if m <= pos:
            return None
        segs.append((pos, m, k * ri, ri))
        pos = m + 2
    if pos >= h.scan_end or any(data.find(r, pos, h.scan_end) >= 0 for r in _RST):
        return None
    segs.append((pos, h.scan_end, (nseg - 1) * ri, mcus - (nseg - 1) * ri))
    return segs


def segments_from_markers(pos, host_bytes, lo, hi, nseg, good):
    """Restart intervals of several scans from the sorted positions `pos` of every 0xFF 0xDk pair in `host_bytes`
    (the device's search, cama_jpeg_find_restarts): scan j occupies bytes [lo[j], hi[j]) and must hold nseg[j]
    intervals.  Clears good[j] (in place) for a scan whose markers are not exactly nseg[j] - 1, do not cycle
    RST0..RST7, or leave an interval empty (T.81 B.2.1, E.2.4): those go to the host decoder.  Returns (g, ns, k, starts,
    ends): the surviving scans, their interval counts, and per interval its number inside the scan and its byte range.
    Same segmentation as restart_segments(), for a whole group at once."""
    A = np.searchsorted(pos, lo, "left")
    B = np.searchsorted(pos, hi - 1, "left")                       # a marker is two bytes: pos + 2 <= hi
    good &= (B - A) == nseg - 1
    for _ in range(2):                                             # second pass: without the scans the checks dropped
        g = np.flatnonzero(good)
        ns = nseg[g]
        total = int(ns.sum())
        first = np.cumsum(ns) - ns                                 # index of every scan's first interval
        M = np.concatenate([pos[a:b] for a, b in zip(A[g], B[g])]) if len(g) else np.zeros(0, np.int64)
        k = np.arange(total) - np.repeat(first, ns)                # interval number inside its scan
        starts, ends = np.empty(total, np.int64), np.empty(total, np.int64)
        starts[first] = lo[g]
        ends[first + ns - 1] = hi[g]
        not_first, not_last = k > 0, k < np.repeat(ns, ns) - 1
        starts[not_first] = M + 2
        ends[not_last] = M
        bad = ends <= starts
        bad[not_last] |= (host_bytes[M + 1] & 7) != (k[not_last] & 7)
        if not bad.any():
            break
        good[g[np.add.reduceat(bad.astype(np.int64), first) > 0]] = False
    return g, ns, k, starts, ends


def _host_decode(data, bgr):
    from PIL import Image
    with Image.open(io.BytesIO(data)) as im:
        arr = np.array(im.convert("RGB"))
    return arr[:, :, ::-1] if bgr else arr


class DeviceJpegDecoder:
    """Batch decoder bound to one GPU.  decode(list of JPEG byte strings) -> uint8 tensor [n, H, W, 3] on the device
    (all images of a batch must share one size; BGR by default = cv2.imread's order)."""

    # decode workgroup: 256 subsequences of 1024 bits (jpeg_kernels.hpp: JPEG_WG, JPEG_SUB_BITS), 47 KB of LDS = 3 per CU
    WG_BYTES, WGS_PER_CU = 256 * 1024 // 8, 3

    def __init__(self, device, lanes=None, min_group=8):
        import torch
        self.lib = _lib.lib()
        if not torch.cuda.is_available():
            raise _lib.CamaHipError("DeviceJpegDecoder needs a GPU; there is no CPU fallback for the device path")
        assert self.lib.cama_jpeg_image_bytes() == IMAGE_DTYPE.itemsize
        assert self.lib.cama_jpeg_huff_set_bytes() == HUFF_DTYPE.itemsize
        self.device = torch.device(device)
        self._huff_index, self._huff_host, self._huff_dev = {}, [], None
        self._quant_index, self._quant_host, self._quant_dev = {}, [], None
        # The decode kernels last as long as their slowest workgroup (the longest re-synchronisation chain), not as
        # long as their work, so a batch is split into groups of images on separate streams ("lanes", each with its own
        # staging buffer and scratch) whose kernels run side by side.  How many: a group whose entropy stages need about
        # HALF of the chip's decode-workgroup slots (3 per CU: LDS) lets two groups tile the chip while the next ones
        # upload -- 240 photo-like 1600x900 frames (10 workgroups each): 7 groups 42-43 k images/s, 4 groups 36 k, 8 groups
        # 37 k, 1 group 33 k (profiles/r02_jpeg_lanes.txt).  `lanes` fixes the number of groups instead (A/B).
        self.lanes = None if lanes is None else int(lanes)
        self.min_group = int(min_group)
        self.group_wgs = self.WGS_PER_CU * torch.cuda.get_device_properties(self.device).multi_processor_count // 2
        self.max_lanes = 16
        self._copy_stream = None
        import threading
        self._lock = threading.Lock()      # lanes + stats: decode_async may run on a pump thread, result() on the consumer's
        self._lane = []
        self._templates = {}
        self.stats = {"device": 0, "host_unsupported": 0, "host_flagged": 0}

    # ------------------------------------------------------------------ pinned arenas for the readers
    def arena(self, nbytes):
        """A PinnedArena of at least `nbytes`: recycled from the pool when nobody references an old one any more (blobs,
        pending decodes and tickets hold references; a ticket lives until its decode has been waited for)."""
        import sys
        pool = self.__dict__.setdefault("_arenas", [])
        for a in pool:
            if a.size >= nbytes and sys.getrefcount(a) <= 3:       # the pool's list, `a`, getrefcount's argument
                a.used = 0
                return a
        a = PinnedArena(max(int(nbytes), 64 << 20))
        pool.append(a)
        if len(pool) > 8:
            pool[:] = [x for x in pool if sys.getrefcount(x) > 3 or x is a][-8:]
        return a

    def stage(self, datas):
        """bytes-like objects -> ArenaBlobs in one arena (for callers that hold files in memory; readers that can should
        read straight into `arena(...).take(n).view()`)."""
        total = sum(len(d) + 16 for d in datas)
        a = self.arena(total)
        out = []
        for d in datas:
            b = a.take(len(d))
            b.view()[:] = np.frombuffer(d, np.uint8)
            out.append(b)
        return out

    # ------------------------------------------------------------------ table caches (device copies grow on demand)
    def _huff_id(self, huff):
        key = tuple(sorted((k, v) for k, v in huff.items() if k[1] <= 1))
        idx = self._huff_index.get(key)
        if idx is None:
            idx = self._huff_index[key] = len(self._huff_host)
            self._huff_host.append(build_huff_set(huff))
            self._huff_dev = None
        return idx

    def _quant_id(self, quant):
        key = quant.tobytes()
        idx = self._quant_index.get(key)
        if idx is None:
            idx = self._quant_index[key] = len(self._quant_host)
            self._quant_host.append(quant.copy())
            self._quant_dev = None
        return idx

    def _template(self, h):
        """Descriptor with everything that comes from the header filled in; cached per parsed header (the frames of a
        camera share one: parse_header hands out copies that reference the same table objects)."""
        key = (id(h.huff), id(h.quant), h.width, h.height)
        hit = self._templates.get(key)
        if hit is None or hit[0] is not h.huff:
            rec = np.zeros(1, IMAGE_DTYPE)
            rec["width"], rec["height"], rec["ncomp"], rec["hs"], rec["vs"] = h.width, h.height, h.ncomp, h.hs, h.vs
            rec["huff_set"], rec["quant_set"] = self._huff_id(h.huff), self._quant_id(h.quant)
            rec["comp_dc"], rec["comp_ac"] = h.comp_dc, h.comp_ac
            if len(self._templates) > 256:
                self._templates.clear()
            hit = self._templates[key] = (h.huff, rec)
        return hit[1]

    def _tables(self):
        import torch
        if self._huff_dev is None:
            blob = np.stack(self._huff_host).view(np.uint8).reshape(len(self._huff_host), -1)
            self._huff_dev = torch.from_numpy(blob.copy()).to(self.device)
        if self._quant_dev is None:
            self._quant_dev = torch.from_numpy(np.stack(self._quant_host).astype(np.uint16).view(np.int16)).to(self.device)
        return self._huff_dev, self._quant_dev

    # ------------------------------------------------------------------ decode
    def decode(self, blobs, bgr=True, out=None):
        return self.decode_async(blobs, bgr=bgr, out=out).result()

    def decode_async(self, blobs, bgr=True, out=None, groups=None):
        """Parse, upload and launch; returns a PendingDecode whose result() waits, re-decodes flagged / unsupported
        images on the host and returns the [n, H, W, 3] tensor.  Several decodes may be in flight (each takes its
        own lanes), e.g. the next frame's while the current one is consumed.  `groups` fixes how many groups (streams) THIS
        batch is split into: a caller that keeps several batches in flight anyway (the decode pump: three) passes 1 --
        the batches tile the chip, and a batch's images then go through ONE chain of kernels instead of three shorter,
        staggered ones (main.py loop, 96-image batches: 3.4-3.5 k -> 3.9-4.35 k frames/s)."""
        import torch
        n = len(blobs)
        assert n >= 1
        arena = blobs[0].arena if isinstance(blobs[0], ArenaBlob) else None
        if arena is not None and n >= 2 and all(isinstance(b, ArenaBlob) and b.arena is arena for b in blobs) and \
                all(blobs[k].off + blobs[k].n <= blobs[k + 1].off for k in range(n - 1)):
            return self._decode_arena(blobs, arena, bgr, out, groups)
        headers = [self._header_of(b) for b in blobs]
        size = None
        for h in headers:
            if h is not None:
                size = (h.height, h.width)
                break
        if size is None:                                             # nothing for the device: all on the host
            size = _host_decode(as_bytes(blobs[0]), bgr).shape[:2]
        H, W = size
        ok = [i for i, h in enumerate(headers) if h is not None and (h.height, h.width) == (H, W)]
        with torch.cuda.device(self.device):
            if out is None:
                out = torch.empty((n, H, W, 3), dtype=torch.uint8, device=self.device)
            assert tuple(out.shape) == (n, H, W, 3) and out.is_contiguous() and out.dtype == torch.uint8
            tickets = []
            if ok:
                # decode workgroups per image (from the stuffed length: a slight over-estimate), groups of ~group_wgs
                wgs = np.array([-(-(headers[i].scan_end - headers[i].scan_start) // self.WG_BYTES) for i in ok])
                bounds = self._group_bounds(wgs, groups)
                cur = torch.cuda.current_stream(self.device)
                tickets = [self._submit([blobs[i] for i in ok[lo:hi]], [headers[i] for i in ok[lo:hi]], ok[lo:hi], out, bgr, cur)
                           for lo, hi in zip(bounds[:-1], bounds[1:])]
        return PendingDecode(self, blobs, ok, tickets, out, bgr)

    @staticmethod
    def _header_of(b):
        try:
            return parse_header_blob(b) if isinstance(b, ArenaBlob) else parse_header(b)
        except Unsupported:
            return None

    def _group_bounds(self, wgs, groups):
        """Split images with `wgs` decode workgroups each into groups: equal shares of the workgroups, not of the images
        (filling every group to group_wgs and leaving a small last one measures the same)."""
        cum = np.concatenate([[0], np.cumsum(wgs)])
        n = len(wgs)
        if groups is not None:
            groups = max(1, min(int(groups), max(1, n // self.min_group)))
        elif self.lanes is not None:
            groups = max(1, min(self.lanes, n // self.min_group))
        else:
            groups = max(1, min(self.max_lanes, n // self.min_group, -(-int(cum[-1]) // self.group_wgs)))
        bounds = [0] + [int(np.searchsorted(cum, cum[-1] * g / groups, "left")) for g in range(1, groups)] + [n]
        return sorted(set(bounds))

    def _decode_arena(self, blobs, arena, bgr, out, groups):
        """decode_async for files that sit in ONE pinned arena in order (the pipeline's path).  Nothing about a file has to be
        known to start moving it: the groups are cut by FILE size, every group's span goes to the device at once on the
        decoder's copy stream (back to back, an event each), and the headers are parsed group by group while the bytes
        travel -- the first kernels start after one group's parse instead of after the whole batch's (0.5 ms of a 5 ms batch
        of 240), and a group's upload no longer queues behind the kernels of the group that used its stream before."""
        import torch
        n = len(blobs)
        bounds = self._group_bounds(np.array([-(-b.n // self.WG_BYTES) for b in blobs]), groups)
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(device=self.device)
            cs = self._copy_stream
            span0 = blobs[0].off
            uploads = []
            with torch.cuda.stream(cs):
                dev = torch.empty(blobs[-1].off + blobs[-1].n - span0 + 16, dtype=torch.uint8, device=self.device)
                for lo, hi in zip(bounds[:-1], bounds[1:]):
                    a, b = blobs[lo].off, blobs[hi - 1].off + blobs[hi - 1].n
                    dev[a - span0:b - span0].copy_(arena.buf[a:b], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(cs)
                    a16 = (a - span0) & ~15                           # (cama_jpeg_find_restarts wants a 16-byte aligned start;
                    uploads.append((dev[a16:b - span0], span0 + a16, ev))   # the bytes before `a` are nobody's scan)
            tickets, ok, size, pending = [], [], None, []
            for g, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
                headers = [self._header_of(b) for b in blobs[lo:hi]]
                if size is None:
                    for h in headers:
                        if h is not None:
                            size = (h.height, h.width)
                            break
                    if size is None:
                        continue                                     # (nothing for the device in this group so far)
                    if out is None:
                        out = torch.empty((n,) + size + (3,), dtype=torch.uint8, device=self.device)
                    assert tuple(out.shape) == (n,) + size + (3,) and out.is_contiguous() and out.dtype == torch.uint8
                sel = [k for k, h in enumerate(headers) if h is not None and (h.height, h.width) == size]
                if sel:
                    slots = [lo + k for k in sel]
                    ok += slots
                    tickets.append(self._submit([blobs[i] for i in slots], [headers[k] for k in sel], slots, out, bgr, cur,
                                                uploaded=uploads[g]))
            if size is None:                                         # nothing for the device: all on the host
                size = _host_decode(as_bytes(blobs[0]), bgr).shape[:2]
                if out is None:
                    out = torch.empty((n,) + tuple(size) + (3,), dtype=torch.uint8, device=self.device)
        return PendingDecode(self, blobs, ok, tickets, out, bgr, keep=(dev, arena))

    def _lane_acquire(self):
        import torch
        with self._lock:
            for L in self._lane:
                if L is not None and not L["busy"]:
                    L["busy"] = True
                    return L
            L = {"stream": torch.cuda.Stream(device=self.device), "pinned": None, "scratch": None, "status": None,
                 "desc": None, "busy": True}
            self._lane = [x for x in self._lane if x is not None] + [L]
            return L

    @staticmethod
    def _repeat_templates(tmpl, counts):
        """Descriptor block: template k repeated counts[k] times."""
        if all(t is tmpl[0] for t in tmpl):                      # one camera model: the usual case
            return np.repeat(tmpl[0], int(counts.sum()))
        blk = np.empty(int(counts.sum()), IMAGE_DTYPE)           # (np.concatenate on structured arrays is slow)
        at = 0
        for t, c in zip(tmpl, counts.tolist()):
            blk[at:at + c] = t[0]
            at += c
        return blk

    def _restart_descriptors(self, L, stream_dev, stream_bytes, host_bytes, headers, dri, base, lens, nd0):
        """Descriptors of the restart-interval images `dri` of a group: (parents [len(dri)], segments, the image of every
        segment, images to hand to the host decoder).  The RSTn markers are located by the device in the bytes already
        uploaded (one small readback per group); an image whose markers do not cycle RST0..RST7 in the number its frame
        needs, or with an empty interval, is left to the host decoder (T.81 B.2.1, E.2.4)."""
        import torch
        H = [headers[i] for i in dri]
        lo = base[dri].astype(np.int64)
        hi = lo + lens[dri]
        mcus = np.array([-(-h.width // (8 * h.hs)) * -(-h.height // (8 * h.vs)) for h in H], dtype=np.int64)
        ri = np.array([h.restart_interval for h in H], dtype=np.int64)
        bpm = np.array([1 if h.ncomp == 1 else h.hs * h.vs + 2 for h in H], dtype=np.int64)
        nseg = -(-mcus // ri)
        # the descriptor array holds at most 65535 entries: images past that budget go to the host decoder
        good = nd0 + len(dri) + np.cumsum(nseg) <= 65535
        capacity = int(nseg[good].sum()) + 4096                    # + strays in file headers inside the uploaded span
        st = L["stream"]
        with torch.cuda.stream(st):
            if L.get("rst") is None or L["rst"][0].numel() < capacity + 1:
                L["rst"] = (torch.empty(capacity * 5 // 4 + 1, dtype=torch.int32, device=self.device),
                            torch.empty(capacity * 5 // 4 + 1, dtype=torch.int32).pin_memory())
            dev, pin = L["rst"]
            _lib.check(self.lib.cama_jpeg_find_restarts(stream_dev.data_ptr(), stream_bytes, dev.data_ptr() + 4,
                                                        capacity, dev.data_ptr(), st.cuda_stream))
            pin[:capacity + 1].copy_(dev[:capacity + 1], non_blocking=True)
        st.synchronize()
        got = pin.numpy().view(np.uint32)
        found = int(got[0])
        if found > capacity:                                       # not JPEG entropy data: nothing to trust
            good[:] = False
            found = 0
        pos = np.sort(got[1:1 + found].astype(np.int64))
        g, ns, k, starts, ends = segments_from_markers(pos, host_bytes, lo, hi, nseg, good)
        tmpl = [self._template(h) for h in H]
        parents = self._repeat_templates(tmpl, np.ones(len(dri), np.int64))
        parents["kind"], parents["out_slot"] = KIND_PIXELS, dri
        if not len(g):
            return parents, np.empty(0, IMAGE_DTYPE), np.zeros(0, np.int64), [int(i) for i in dri]
        segs = self._repeat_templates([tmpl[j] for j in g], ns)
        rg = np.repeat(np.arange(len(g)), ns)
        segs["kind"], segs["parent"], segs["out_slot"] = KIND_SEGMENT, nd0 + np.repeat(g, ns), 0
        segs["first_block"] = k * ri[g][rg] * bpm[g][rg]
        nmcu = np.minimum(ri[g][rg], mcus[g][rg] - k * ri[g][rg])
        segs["width"] = nmcu * 8 * np.array([H[j].hs for j in g], dtype=np.int64)[rg]
        segs["height"] = 8 * np.array([H[j].vs for j in g], dtype=np.int64)[rg]
        segs["stream_off"], segs["stream_len"] = starts, ends - starts
        dri = np.asarray(dri)
        return parents, segs, dri[g][rg], [int(i) for i in dri[~good]]

    def _submit(self, blobs, headers, slots, out, bgr, cur, uploaded=None):
        """Pack, upload and launch the decode of one group on a free lane's stream; returns a ticket for _finish."""
        import torch
        L = self._lane_acquire()
        n = len(blobs)
        # descriptors: one per image without restart intervals; with them, a pixels-only parent + one entropy segment
        # per interval (include/cama_hip.h)
        parts, owner, broken = [], [], []
        nd = 0
        # the scans go into the staging buffer back to back (no alignment needed: descriptors point into it, the
        # library lays out its own aligned unstuffed copies)
        lens = np.array([h.scan_end - h.scan_start for h in headers], dtype=np.int64)
        arena = blobs[0].arena if isinstance(blobs[0], ArenaBlob) else None
        if uploaded is not None:
            # the group's span is already on its way (decode_async's copy stream): descriptors point into it
            stream_dev, span_lo, upload_done = uploaded
            span_hi = span_lo + stream_dev.numel()
            base = np.array([b.off - span_lo + h.scan_start for b, h in zip(blobs, headers)], dtype=np.int64)
            stream_bytes = int(span_hi - span_lo)
        elif arena is not None and all(isinstance(b, ArenaBlob) and b.arena is arena for b in blobs) and \
                all(blobs[k].off < blobs[k + 1].off for k in range(n - 1)):
            # the files already sit in pinned memory, in order: upload the span they occupy as it is (headers included,
            # ~0.2 % of the bytes) and point the descriptors into it -- no packing copy
            span_lo, span_hi = blobs[0].off, blobs[-1].off + blobs[-1].n
            base = np.array([b.off - span_lo + h.scan_start for b, h in zip(blobs, headers)], dtype=np.int64)
            stream_bytes = int(span_hi - span_lo)
            pinned_src = arena.buf[span_lo:span_hi]
        else:
            arena = None
            base = np.concatenate([[0], np.cumsum(lens)[:-1]])
            stream_bytes = int(lens.sum())
            if L["pinned"] is None or L["pinned"].numel() < stream_bytes:
                # (grown with headroom: lanes are reused across batch sizes, regrowing pinned / device buffers is slow)
                L["pinned"] = torch.empty(max(stream_bytes * 5 // 4, 1 << 22), dtype=torch.uint8).pin_memory()
            host = L["pinned"].numpy()
            for o, ln, b, h in zip(base.tolist(), lens.tolist(), blobs, headers):
                host[o:o + ln] = np.frombuffer(as_bytes(b), np.uint8, ln, h.scan_start)
            pinned_src = L["pinned"][:stream_bytes]
        host_bytes = arena.np[span_lo:span_hi] if arena is not None else host[:stream_bytes]
        st = L["stream"]
        st.wait_stream(cur)                                # `out`, the tables and earlier work of the caller
        if uploaded is not None:
            st.wait_event(upload_done)
        else:
            with torch.cuda.stream(st):
                stream_dev = pinned_src.to(self.device, non_blocking=True)
        # descriptors: images without restart intervals first, as one vectorised block
        whole = [i for i, h in enumerate(headers) if not h.restart_interval]
        if whole:
            tmpl = [self._template(headers[i]) for i in whole]
            blk = self._repeat_templates(tmpl, np.ones(len(whole), np.int64))
            blk["kind"], blk["out_slot"], blk["stream_off"], blk["stream_len"] = KIND_WHOLE, whole, base[whole], lens[whole]
            parts.append(blk)
            owner.append(np.asarray(whole))
            nd = len(whole)
        # restart intervals: a pixels-only parent + one entropy segment per interval, all pointing into the one upload;
        # the markers between them are found on the device (cama_jpeg_find_restarts)
        dri = [i for i, h in enumerate(headers) if h.restart_interval]
        if dri:
            parents, segs, seg_owner, broken = self._restart_descriptors(L, stream_dev, stream_bytes, host_bytes, headers,
                                                                         dri, base, lens, nd)
            parts += [parents, segs]
            owner += [np.asarray(dri), seg_owner]
            nd += len(parents) + len(segs)
        if len(parts) == 1:
            imgs = parts[0]
        else:
            imgs = np.empty(nd, IMAGE_DTYPE)
            at = 0
            for part in parts:
                imgs[at:at + len(part)] = part
                at += len(part)
        owner = np.concatenate(owner)
        info = np.zeros(3, np.uint64)                      # cama_jpeg_plan_info: u64 scratch_bytes + 4 x u32
        _lib.check(self.lib.cama_jpeg_plan(imgs.ctypes.data, nd, stream_bytes, info.ctypes.data))
        scratch_bytes = int(info[0])
        huff_dev, quant_dev = self._tables()               # (uploaded on the caller's stream when a new table set appeared)
        st.wait_stream(cur)
        contiguous = slots == list(range(slots[0], slots[0] + n))
        with torch.cuda.stream(st):
            if L["scratch"] is None or L["scratch"].numel() < scratch_bytes:
                L["scratch"] = None
                L["scratch"] = torch.empty(scratch_bytes * 5 // 4, dtype=torch.uint8, device=self.device)
            # (descriptors through a pinned buffer of the lane: a copy from pageable memory is staged synchronously)
            raw = imgs.view(np.uint8).reshape(-1)
            if L["desc"] is None or L["desc"].numel() < raw.size:
                L["desc"] = torch.empty(max(raw.size * 2, 1 << 14), dtype=torch.uint8).pin_memory()
            L["desc"][:raw.size].numpy()[:] = raw
            imgs_dev = L["desc"][:raw.size].to(self.device, non_blocking=True).view(nd, -1)
            status = torch.empty(nd, dtype=torch.int32, device=self.device)
            target = out[slots[0]:slots[0] + n] if contiguous else \
                torch.empty((n,) + tuple(out.shape[1:]), dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.cama_jpeg_decode(
                stream_dev.data_ptr(), stream_bytes, imgs.ctypes.data, imgs_dev.data_ptr(), nd, huff_dev.data_ptr(),
                huff_dev.shape[0], quant_dev.data_ptr(), quant_dev.shape[0], target.data_ptr(), target.stride(0),
                int(bool(bgr)), L["scratch"].data_ptr(), L["scratch"].numel(), status.data_ptr(), st.cuda_stream))
            if not contiguous:
                out[torch.as_tensor(slots, device=self.device)] = target
            if L["status"] is None or L["status"].numel() < nd:
                L["status"] = torch.empty(max(nd, 256), dtype=torch.int32).pin_memory()
            L["status"][:nd].copy_(status, non_blocking=True)
        # (no record_stream on `out`: _finish synchronises every lane before decode() returns)
        return {"lane": L, "slots": slots, "n": nd, "owner": owner, "broken": broken,
                "keep": (stream_dev, imgs_dev, status, target, imgs, arena)}

    def _finish(self, ticket, out, cur):
        """Wait for a group; returns the slots the device flagged as inconsistent (to be decoded on the host)."""
        st = ticket["lane"]["stream"]
        st.synchronize()                                   # one small readback per group (also frees the staging buffer)
        cur.wait_stream(st)
        bad = ticket["lane"]["status"][:ticket["n"]].numpy().copy()
        with self._lock:
            ticket["lane"]["busy"] = False
        local = {int(ticket["owner"][r]) for r in np.flatnonzero(bad)} | set(ticket["broken"])   # descriptor -> image
        return [ticket["slots"][i] for i in sorted(local)]


class PendingDecode:
    def __init__(self, dec, blobs, ok, tickets, out, bgr, keep=None):
        self.dec, self.blobs, self.ok, self.tickets, self.out, self.bgr = dec, blobs, ok, tickets, out, bgr
        self._keep = keep                                  # (the batch's device copy of the arena span, until result())
        self._done = False

    def result(self):
        import torch
        if self._done:
            return self.out
        dec, out, n = self.dec, self.out, len(self.blobs)
        with torch.cuda.device(dec.device):
            cur = torch.cuda.current_stream(dec.device)
            flagged = []
            for t in self.tickets:
                flagged += dec._finish(t, out, cur)
            on_device = set(self.ok)
            host = [i for i in range(n) if i not in on_device] + flagged
            with dec._lock:
                dec.stats["device"] += len(self.ok) - len(flagged)
                dec.stats["host_unsupported"] += n - len(self.ok)
                dec.stats["host_flagged"] += len(flagged)
            for i in host:
                arr = _host_decode(as_bytes(self.blobs[i]), self.bgr)
                if tuple(arr.shape[:2]) != tuple(out.shape[1:3]):
                    raise ValueError(f"image {i} is {arr.shape[1]}x{arr.shape[0]}, the batch is {out.shape[2]}x{out.shape[1]}")
                out[i].copy_(torch.from_numpy(np.ascontiguousarray(arr)))
            # `out` may have been allocated under another stream than the consumer's (the decode pump's own): the allocator
            # must not hand the block out again while the consumer's stream still reads it
            out.record_stream(cur)
        self._done = True
        self.blobs = self.tickets = self._keep = None
        return out
