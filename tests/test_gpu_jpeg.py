"""Device JPEG decoder (cama_jpeg_decode through cama_amd.jpeg.DeviceJpegDecoder) against the real decoder: Pillow's
bundled libjpeg-turbo, byte for byte -- sizes off the MCU grid, all supported subsamplings, qualities, grey, optimised
Huffman tables, mixed batches, host fallback for files outside the device scope, and a full-size 1600x900 rig."""
import io
import os

import numpy as np
import pytest
from PIL import Image

from tests.test_oracle_jpeg import encode, pillow_rgb, synth_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dec():
    import torch
    assert torch.cuda.is_available()
    from cama_amd.jpeg import DeviceJpegDecoder
    return DeviceJpegDecoder("cuda:0")


@pytest.mark.parametrize("size", [(16, 16), (8, 8), (17, 23), (48, 64), (90, 160), (1, 1), (15, 31), (203, 317),
                                  (21, 3), (5, 4), (33, 2), (2, 5), (3, 6)])
@pytest.mark.parametrize("sub", [0, 1, 2])
def test_device_decode_equals_libjpeg_turbo(dec, size, sub):
    blobs = [encode(synth_image(size[0], size[1], kind, seed=q), quality=q, subsampling=sub)
             for kind in ("smooth", "noise", "edges") for q in (50, 90, 100)]
    before = dict(dec.stats)
    got = dec.decode(blobs, bgr=False).cpu().numpy()
    assert dec.stats["device"] - before["device"] == len(blobs), dec.stats        # none fell back to the host
    for k, b in enumerate(blobs):
        assert np.array_equal(got[k], pillow_rgb(b)), (size, sub, k)
    bgr = dec.decode(blobs[:2]).cpu().numpy()
    assert np.array_equal(bgr[0], pillow_rgb(blobs[0])[:, :, ::-1])


def test_grey_optimised_tables_and_mixed_batch(dec):
    img = synth_image(120, 200, "noise", seed=5)
    img[:, :100] = synth_image(120, 100, "smooth")
    blobs = [encode(img[..., 0], quality=85), encode(img, quality=80, subsampling=2, optimize=True),
             encode(img, quality=95, subsampling=0), encode(img, quality=30, subsampling=1, optimize=True)]
    got = dec.decode(blobs, bgr=False).cpu().numpy()
    for k, b in enumerate(blobs):
        assert np.array_equal(got[k], pillow_rgb(b)), k


def test_table_sets_without_a_second_level_decode_the_same(monkeypatch):
    """The codes of 11..16 bits come from a second-level table per set; a set whose long codes do not fit it is marked
    (l2_off = L2_NONE) and walks the per-length limits instead.  Forced here for every table: same bytes, nothing flagged.
    Noise at quality 100 uses the 16-bit codes all the time."""
    from cama_amd import jpeg as PJ

    def no_second_level(rec):
        rec["l2_first"][:] = 0
        rec["l2_off"][:] = PJ.L2_NONE

    monkeypatch.setattr(PJ, "_second_level", no_second_level)
    dec = PJ.DeviceJpegDecoder("cuda:0")
    img = synth_image(203, 317, "noise", seed=9)
    img[:, :150] = synth_image(203, 150, "smooth")
    blobs = [encode(img, quality=q, subsampling=sub, **kw) for q in (35, 90, 100) for sub in (0, 2)
             for kw in ((dict(), dict(optimize=True)) if sub else (dict(),))]  # (Pillow cannot optimise 4:4:4 noise this size)
    got = dec.decode(blobs, bgr=False).cpu().numpy()
    assert dec.stats["device"] == len(blobs) and not dec.stats["host_flagged"], dec.stats
    for k, b in enumerate(blobs):
        assert np.array_equal(got[k], pillow_rgb(b)), k


def test_restart_intervals_decode_on_the_device(dec):
    """DRI files: every restart interval is its own entropy segment (exact start state, DC predictors reset)."""
    rng = np.random.default_rng(4)
    blobs = []
    for (h, w) in [(64, 96), (203, 317), (33, 47)]:
        img = synth_image(h, w, "noise", seed=h)
        img[:, :w // 2] = synth_image(h, w // 2, "smooth")
        for sub in (0, 1, 2):
            for kw in (dict(restart_marker_blocks=1), dict(restart_marker_blocks=5), dict(restart_marker_rows=1),
                       dict(restart_marker_rows=2, optimize=True)):
                blobs.append((h, w, encode(img, quality=int(rng.integers(30, 100)), subsampling=sub, **kw)))
    before = dict(dec.stats)
    for (h, w) in [(64, 96), (203, 317), (33, 47)]:
        group = [b for (hh, ww, b) in blobs if (hh, ww) == (h, w)] + [encode(synth_image(h, w, "edges"), quality=80)]
        got = dec.decode(group, bgr=False).cpu().numpy()
        for k, b in enumerate(group):
            assert np.array_equal(got[k], pillow_rgb(b)), (h, w, k)
    assert dec.stats["device"] - before["device"] == len(blobs) + 3 and dec.stats["host_flagged"] == before["host_flagged"]
    big = encode(synth_image(900, 1600, "noise", seed=1), quality=90, restart_marker_rows=1)
    got = dec.decode([big], bgr=False).cpu().numpy()
    assert np.array_equal(got[0], pillow_rgb(big)) and dec.stats["host_flagged"] == before["host_flagged"]
    # long intervals (a noise frame with a marker every 4 MCU rows: ~90 KB each, several decode workgroups per interval)
    # mixed with short-interval and marker-free files
    from cama_amd.jpeg import parse_header, restart_segments
    long_iv = encode(synth_image(900, 1600, "noise", seed=2), quality=92, restart_marker_rows=4)
    assert max(e - s for s, e, _, _ in restart_segments(long_iv, parse_header(long_iv))) > (1 << 16)
    group = [long_iv, big, encode(synth_image(900, 1600, "smooth"), quality=85),
             encode(synth_image(900, 1600, "edges"), quality=70, restart_marker_blocks=3, subsampling=0)]
    got = dec.decode(group, bgr=False).cpu().numpy()
    for k, b in enumerate(group):
        assert np.array_equal(got[k], pillow_rgb(b)), k
    assert dec.stats["host_flagged"] == before["host_flagged"]


def test_damaged_restart_interval_is_flagged(dec):
    """Bytes changed INSIDE one restart interval (markers intact): that interval ends with the wrong block count or an
    inconsistent state, the device flags it, and the image is decoded again on the host -- the others stay on the device."""
    from cama_amd.jpeg import parse_header, restart_segments
    rng = np.random.default_rng(8)
    files = [encode(synth_image(120, 200, "noise", seed=k), quality=90, restart_marker_rows=1, subsampling=k % 3) for k in range(6)]
    data = bytearray(files[2])
    segs = restart_segments(files[2], parse_header(files[2]))
    s0, e0 = segs[3][0], segs[3][1]
    for p in range(s0 + 4, min(e0 - 2, s0 + 40)):
        b = int(rng.integers(0, 255))
        data[p] = b if b != 0xFF else 0x7F                       # (no new markers, no new stuffing)
    files[2] = bytes(data)
    before = dict(dec.stats)
    try:
        got = dec.decode(files, bgr=False).cpu().numpy()
    except Exception:
        got = None                                               # Pillow may refuse the damaged file
    assert dec.stats["host_flagged"] - before["host_flagged"] == 1, dec.stats
    if got is not None:
        for k, b in enumerate(files):
            if k != 2:
                assert np.array_equal(got[k], pillow_rgb(b)), k


def test_device_restart_marker_search_equals_numpy(dec):
    """cama_jpeg_find_restarts == a numpy search for every 0xFF 0xD0..0xD7 pair: random bytes (1/256 of them 0xFF),
    markers planted across the 16-byte thread, 1 KB wave and 4 KB workgroup boundaries and at both ends, ragged
    lengths, and a capacity smaller than the number found (count still exact, nothing written past the capacity)."""
    import torch
    from cama_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    for n in (1, 2, 15, 16, 17, 4095, 4096, 4097, 123457, 3_000_001):
        buf = rng.integers(0, 256, n, dtype=np.uint8)
        for at in (0, 14, 15, 16, 1023, 1024, 4094, 4095, 4096, n - 2, n - 1):
            if 0 <= at < n:
                buf[at] = 0xFF
                if at + 1 < n:
                    buf[at + 1] = 0xD0 + (at & 7)
        idx = np.flatnonzero(buf[:-1] == 0xFF)
        want = idx[(buf[idx + 1] & 0xF8) == 0xD0]
        dev = torch.from_numpy(buf).cuda()
        for cap in (len(want) + 8, max(len(want) // 2, 1)):
            out = torch.full((cap + 9,), -1, dtype=torch.int32, device="cuda")
            _lib.check(L.cama_jpeg_find_restarts(dev.data_ptr(), n, out.data_ptr() + 4, cap, out.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream))
            got = out.cpu().numpy()
            assert got[0] == len(want), (n, cap)
            stored = got[1:1 + min(cap, len(want))]
            assert (got[1 + cap:] == -1).all()
            if cap >= len(want):
                assert np.array_equal(np.sort(stored), want), (n, cap)
            else:
                assert len(set(stored.tolist())) == cap and set(stored.tolist()) <= set(want.tolist())
    assert L.cama_jpeg_find_restarts(dev.data_ptr() + 1, 100, out.data_ptr() + 4, 4, out.data_ptr(), 0) == -1   # alignment


def test_restart_files_with_broken_marker_sequences_go_to_the_host(dec):
    """A group of DRI files, staged in an arena (headers inside the uploaded span), four of them with a missing / out of
    cycle / doubled / surplus RSTn: exactly those are handed to the host decoder, every image equals Pillow's decode."""
    from cama_amd import jpeg as PJ
    rng = np.random.default_rng(21)
    files = []
    for k in range(24):
        img = synth_image(120, 200, "noise", seed=k)
        img[:, :100] = synth_image(120, 100, "smooth")
        kw = [dict(restart_marker_blocks=1), dict(restart_marker_blocks=7), dict(restart_marker_rows=1), {}][k % 4]
        files.append(encode(img, quality=int(rng.integers(40, 98)), subsampling=k % 3, **kw))
    broken = {}
    for kind, j in (("missing", 1), ("cycle", 5), ("empty", 9), ("extra", 13)):
        data, hd = bytearray(files[j]), PJ.parse_header(files[j])
        segs = PJ.restart_segments(bytes(data), hd)
        m = segs[1][1]
        if kind == "missing":
            data[m:m + 2] = b"\x12\x34"
        elif kind == "cycle":
            data[m + 1] = 0xD0 | ((data[m + 1] + 3) & 7)
        elif kind == "empty":
            data[m + 2:m + 2] = bytes((0xFF, 0xD0 | ((data[m + 1] + 1) & 7)))
        else:
            data[segs[-1][0] + 1:segs[-1][0] + 1] = b"\xff\xd3"
        files[j] = bytes(data)
        broken[j] = kind
    want = []
    for j, f in enumerate(files):
        try:
            want.append(pillow_rgb(f))
        except Exception:
            want.append(None)
    assert all(w is not None for j, w in enumerate(want) if j not in broken)
    for blobs in (dec.stage(files), files):
        before = dict(dec.stats)
        try:
            got = dec.decode(blobs, bgr=False).cpu().numpy()
        except Exception:
            assert any(want[j] is None for j in broken)            # Pillow refused one of the damaged files
            continue
        assert dec.stats["host_flagged"] - before["host_flagged"] == len(broken), dec.stats
        for j, w in enumerate(want):
            if w is not None:
                assert np.array_equal(got[j], w), (j, broken.get(j))


def test_files_outside_the_device_scope_fall_back_to_the_host(dec):
    img = synth_image(64, 96, "smooth")
    cmyk = io.BytesIO()
    Image.fromarray(img).convert("CMYK").save(cmyk, format="JPEG")
    blobs = [encode(img, quality=90, progressive=True), cmyk.getvalue(), encode(img, quality=90)]
    before = dict(dec.stats)
    got = dec.decode(blobs, bgr=False).cpu().numpy()
    assert dec.stats["host_unsupported"] - before["host_unsupported"] == 2
    for k, b in enumerate(blobs):
        assert np.array_equal(got[k], pillow_rgb(b)), k


def test_corrupt_stream_is_flagged_and_redecoded_on_the_host(dec):
    img = synth_image(96, 128, "noise", seed=9)
    good = encode(img, quality=90)
    from cama_amd.jpeg import parse_header
    h = parse_header(good)
    cut = bytearray(good)
    del cut[h.scan_start + (h.scan_end - h.scan_start) // 2:h.scan_end - 8]       # drop the second half of the scan
    before = dict(dec.stats)
    try:
        got = dec.decode([good, bytes(cut)], bgr=False).cpu().numpy()
    except Exception:
        got = None                                                                # Pillow may refuse the truncated file
    assert dec.stats["host_flagged"] - before["host_flagged"] == 1
    if got is not None:
        assert np.array_equal(got[0], pillow_rgb(good))


def test_full_size_rig(dec):
    """Six 1600x900 frames (one per camera), noise (worst case: ~1.3 MB each) and photo-like content."""
    rng = np.random.default_rng(0)
    y, x = np.mgrid[0:900, 0:1600]
    base = np.stack([(x * 0.16 + 20 * np.sin(y / 30)) % 256, (y * 0.28) % 256, ((x + y) * 0.1) % 256], -1)
    imgs = [rng.integers(0, 256, (900, 1600, 3), dtype=np.uint8) for _ in range(2)] + \
           [np.clip(base + rng.normal(0, s, base.shape), 0, 255).astype(np.uint8) for s in (0, 4, 8, 16)]
    blobs = [encode(im, quality=90) for im in imgs]
    got = dec.decode(blobs).cpu().numpy()
    for k, b in enumerate(blobs):
        assert np.array_equal(got[k][:, :, ::-1], pillow_rgb(b)), k


def test_decode_does_not_depend_on_what_the_scratch_held(dec):
    """The chain has one fill in front of it (the coefficients); the slack behind the unstuffed bytes, the status words and the
    DC differences are written by the kernels themselves.  Poisoned lane scratch (every byte 0xA5 / 0xFF) must give the same
    pixels and no flagged image -- for streams that end anywhere inside a dword, with and without stuffed bytes."""
    import torch
    sets = [[encode(synth_image(h, w, kind, seed=h + w), quality=q, subsampling=sub)
             for kind in ("noise", "edges") for q, sub in ((100, 0), (75, 2))] for (h, w) in ((203, 317), (90, 160), (17, 23))]
    for poison in (0xA5, 0xFF, 0x00):
        for blobs in sets:                                               # (one size per batch)
            dec.decode(blobs[:2], bgr=False)                             # (make sure a lane with scratch exists)
            torch.cuda.synchronize()
            for lane in dec._lane:
                if lane is not None and lane["scratch"] is not None:
                    lane["scratch"].fill_(poison)
            torch.cuda.synchronize()
            before = dict(dec.stats)
            got = dec.decode(blobs, bgr=False).cpu().numpy()
            assert dec.stats["device"] - before["device"] == len(blobs), (poison, dec.stats)
            for k, b in enumerate(blobs):
                assert np.array_equal(got[k], pillow_rgb(b)), (poison, k)


def test_clip_frame_source_device_decode_equals_host_decode(tmp_path):
    """ClipFrameSource (the demo's ingest): JPEG bytes -> device decoder gives the same BGR frames as the host
    decoder threads, for a synthetic clip with 6 cameras."""
    import torch
    from cama_amd import frames as FR
    from cama_amd.dataset import ClipManager
    from cama_amd.synth import DEFAULT_CAMA_CONFIGS, make_clip
    clip = str(tmp_path / "clip")
    make_clip(clip, n_frames=4, seed=2, n_lines=4, verts_per_line=4, line_len_m=2.0, raster_size=300,
              image_mode="jpg", image_size=(180, 320), with_nuscenes=False, extra_labels=False)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)
    dev = FR.ClipFrameSource(cm.cm_list, torch.device("cuda:0"), decoder="device")
    host = FR.ClipFrameSource(cm.cm_list, torch.device("cuda:0"), decoder="host")
    before = dict(dev._decoder().stats)              # (the decoder is the process-wide one, Engine.jpeg_decoder: deltas)
    a = dev.raw_batch([1, 2, 3]).cpu().numpy()
    b = host.raw_batch([1, 2, 3]).cpu().numpy()
    assert a.shape == b.shape == (3, 6, 180, 320, 3) and np.array_equal(a, b)
    from cama_amd import runtime
    assert dev._jpeg is runtime.engine().jpeg_decoder()
    assert dev._jpeg.stats["device"] - before["device"] == 18 and dev._jpeg.stats["host_flagged"] == before["host_flagged"]
    # the device path read the files straight into a pinned arena (no packing copy on the submitting thread)
    from cama_amd.jpeg import ArenaBlob
    blobs = [f.result()[0] for f in dev._submit(2)]
    assert all(isinstance(x, ArenaBlob) for x in blobs) and len({id(x.arena) for x in blobs}) == 1
    assert [x.off for x in blobs] == sorted(x.off for x in blobs)
    for x, cmgr in zip(blobs, cm.cm_list):
        assert x.tobytes() == open(cmgr.get_image_path(2, True), "rb").read()


def test_clip_frame_source_survives_files_that_change_size(tmp_path):
    """The pinned slices are sized from ONE directory scan per camera; a file rewritten with another length afterwards
    (longer: the slice is too short, shorter: a short read) is read again the plain way and still decodes to what is on disk."""
    import torch
    from cama_amd import frames as FR
    from cama_amd.dataset import ClipManager
    from cama_amd.synth import DEFAULT_CAMA_CONFIGS, make_clip
    clip = str(tmp_path / "clip")
    make_clip(clip, n_frames=16, seed=3, n_lines=4, verts_per_line=4, line_len_m=2.0, raster_size=300,
              image_mode="jpg", image_size=(180, 320), with_nuscenes=False, extra_labels=False)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)
    src = FR.ClipFrameSource(cm.cm_list, torch.device("cuda:0"), decoder="device")
    first = src.raw_batch([1]).cpu().numpy()                    # fills the size tables
    assert src._dir_sizes
    img = synth_image(180, 320, "edges", seed=77)
    longer, shorter = cm.cm_list[0].get_image_path(14, True), cm.cm_list[3].get_image_path(14, True)
    open(longer, "wb").write(encode(synth_image(180, 320, "noise", seed=78), quality=100, subsampling=0))
    open(shorter, "wb").write(encode(img, quality=5))
    assert os.path.getsize(longer) > src._file_size(longer) and os.path.getsize(shorter) < src._file_size(shorter)
    got = src.raw_batch([14]).cpu().numpy()                     # (far beyond what reading frame 1 prefetched)
    for c, cmgr in enumerate(cm.cm_list):
        want = pillow_rgb(open(cmgr.get_image_path(14, True), "rb").read())[:, :, ::-1]
        assert np.array_equal(got[0, c], want), c
    assert np.array_equal(src.raw_batch([1]).cpu().numpy(), first)


def test_arena_staged_blobs_decode_like_bytes(dec):
    """decode() of files staged in a pinned arena (span upload, descriptors pointing into it) == decode() of the same
    bytes objects (packed per call) == Pillow; mixed with a file the device hands back to the host decoder, and with a
    batch that straddles two arenas (falls back to the packing path)."""
    import io
    from PIL import Image
    rng = np.random.default_rng(12)
    imgs = [rng.integers(0, 256, (120, 200, 3), dtype=np.uint8) for _ in range(30)]
    blobs = []
    for k, im in enumerate(imgs):
        b = io.BytesIO()
        Image.fromarray(im).save(b, "JPEG", quality=60 + k, progressive=(k == 7))     # one progressive: host fallback
        blobs.append(b.getvalue())
    want = np.stack([np.array(Image.open(io.BytesIO(b)).convert("RGB"))[:, :, ::-1] for b in blobs])
    staged = dec.stage(blobs)
    assert np.array_equal(dec.decode(staged).cpu().numpy(), want)
    assert np.array_equal(dec.decode(blobs).cpu().numpy(), want)
    other = dec.stage(blobs[10:13])                                    # another arena: the batch is no longer one span
    assert other[0].arena is not staged[0].arena
    mixed = staged[:10] + other + staged[13:]
    assert np.array_equal(dec.decode(mixed).cpu().numpy(), want)


def test_fuzz_random_images_sizes_and_encoder_settings(dec):
    """Randomised differential test against libjpeg-turbo: sizes 1..400, content mixes, qualities 1..100, all
    subsamplings, grey, optimised tables.  CAMA_FUZZ_ITERS scales it (default 60 batches of 4)."""
    import os
    rng = np.random.default_rng(int(os.environ.get("CAMA_FUZZ_SEED", "99")))
    iters = int(os.environ.get("CAMA_FUZZ_ITERS", "60"))
    before = dict(dec.stats)
    total = 0
    for it in range(iters):
        h, w = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        blobs = []
        for _ in range(4):
            kind = rng.choice(["smooth", "noise", "edges", "mix", "flat"])
            if kind == "mix":
                img = synth_image(h, w, "smooth", seed=int(rng.integers(1 << 30))).astype(np.int32)
                img = np.clip(img + rng.normal(0, rng.uniform(1, 40), img.shape), 0, 255).astype(np.uint8)
            elif kind == "flat":
                img = np.full((h, w, 3), rng.integers(0, 256, 3), dtype=np.uint8)
            else:
                img = synth_image(h, w, str(kind), seed=int(rng.integers(1 << 30)))
            kw = dict(quality=int(rng.integers(1, 101)), optimize=bool(rng.random() < 0.3))
            r = rng.random()
            if r < 0.15:
                kw["restart_marker_blocks"] = int(rng.integers(1, 40))
            elif r < 0.3:
                kw["restart_marker_rows"] = int(rng.integers(1, 4))
            grey, sub = rng.random() < 0.15, int(rng.integers(0, 3))
            try:
                blobs.append(encode(img[..., 0], **kw) if grey else encode(img, subsampling=sub, **kw))
            except OSError:                          # Pillow's ENCODER gives up on some optimize + size combinations
                kw["optimize"] = False
                kw.pop("restart_marker_blocks", None)
                blobs.append(encode(img[..., 0], **kw) if grey else encode(img, subsampling=sub, **kw))
        got = dec.decode(blobs, bgr=False).cpu().numpy()
        total += len(blobs)
        for k, b in enumerate(blobs):
            assert np.array_equal(got[k], pillow_rgb(b)), (it, k, h, w)
    # Streams that never self-synchronise inside a 256-subsequence workgroup (quality-100 noise: ~450 bits per block)
    # leave a stale workgroup entry; the write pass detects it and the image is re-decoded on the host -- still
    # byte-exact above, but it must stay rare (1 in 3200 with the default seed)
    flagged = dec.stats["host_flagged"] - before["host_flagged"]
    assert dec.stats["host_unsupported"] == before["host_unsupported"]
    assert dec.stats["device"] - before["device"] == total - flagged and flagged <= max(1, total // 400), dec.stats
