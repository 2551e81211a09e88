#!/usr/bin/env python3
"""The reference demo loop (main.py:50-61) on a synthetic clip with real 1600x900 JPEG frames, end to end, with a
per-stage clock: where does a frame's time go once the reprojection hot path runs on the GPU?

    python examples/demo_synthetic.py [--frames 8] [--keep DIR]

The loop itself uses the device JPEG decoder (CAMA_JPEG_DECODER=host switches back to the host thread pool).
Stage clock per frame: JPEG decode on the host (Pillow / cv2) vs on the device, upload (PCIe), device work
(undistort+resize, project, stamp, mosaic: one fused pass), download of the 2880x1080 mosaic.  No encoder (ffmpeg).
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--keep", default=None)
    ap.add_argument("--content", choices=["noise", "photo"], default="photo",
                    help="frame content: photo-like (~300 KB JPEGs) or pure noise (~1.3 MB, the decoder's worst case)")
    args = ap.parse_args()
    import torch
    from cama.dataset import ClipManager               # drop-in import path
    from cama.tools import VideoGenerator
    from cama_amd import frames as FR
    from cama_amd import runtime
    from cama_amd.synth import DEFAULT_CAMA_CONFIGS, make_clip
    root = args.keep or tempfile.mkdtemp(prefix="cama_demo_")
    clip = os.path.join(root, "clip")
    t = time.perf_counter()
    make_clip(clip, n_frames=args.frames + 1, seed=0, n_lines=20, verts_per_line=11, line_len_m=5.0, raster_size=3000,
              image_mode="jpg" if args.content == "noise" else "jpg_photo", image_size=(900, 1600), with_nuscenes=False, extra_labels=False)
    print(f"synthetic clip with {6 * (args.frames + 1)} JPEGs written in {time.perf_counter() - t:.1f} s -> {clip}")
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)          # reference default output size (540, 960)
    vg = object.__new__(VideoGenerator)                          # no encoder
    eng = runtime.engine()
    # 1) the reference loop, verbatim; twice: the first pass pays the one-off setup (library load, static-map upload,
    #    rig, first launches), the second is the steady state
    for label in ("first pass (incl. one-off setup)", "steady state"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for image_idx, instance_map in cm.yield_frame(dataset="cama"):
            maps_2d_dict = cm.project_all_camera(instance_map)
            image_dict = cm.render_vectors(maps_2d_dict, image_idx)
            image = vg.concate_image(image_dict)
            n += 1
        torch.cuda.synchronize()                                 # the mosaics stay on the device here (no encoder)
        dt = time.perf_counter() - t0
        print(f"main.py loop, {label}: {n} frames in {dt:.3f} s = {n / dt:.1f} frames/s, mosaic {image.shape}")
    # 2) stage clock
    idx, w2c = cm.frame_poses("cama")
    t_dec = t_up = t_gpu = t_down = 0.0
    rig = cm._rig()
    dmap = cm._static("cama").device()
    for k, i in enumerate(idx):
        a = time.perf_counter()
        raw = [FR.read_bgr(c.get_image_path(int(i), True)) for c in cm.cm_list]
        b = time.perf_counter()
        dev = torch.from_numpy(np.stack(raw)[None]).to(eng.device)
        torch.cuda.synchronize()
        c_ = time.perf_counter()
        out = eng.render_frames_raw(dmap, rig, w2c[k:k + 1], dev, cm.cm_list)
        torch.cuda.synchronize()
        d = time.perf_counter()
        host = out.cpu().numpy()
        e = time.perf_counter()
        t_dec += b - a; t_up += c_ - b; t_gpu += d - c_; t_down += e - d
    F = len(idx)
    # the same six files through the device decoder (compressed bytes up, decode on the GPU)
    from cama_amd.jpeg import DeviceJpegDecoder
    dec = DeviceJpegDecoder(eng.device)
    blobs = [open(c.get_image_path(int(idx[-1]), True), "rb").read() for c in cm.cm_list]   # the frame `dev` holds
    dev2 = dec.decode(blobs)
    torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(10):
        dec.decode(blobs, out=dev2)
    torch.cuda.synchronize()
    t_devdec = (time.perf_counter() - a) / 10
    assert torch.equal(dev2, dev[0])                     # byte-identical to the host decoder
    print(f"device JPEG decode: 6 frames ({sum(map(len, blobs)) / 6e3:.0f} KB each) in {t_devdec * 1e3:.2f} ms "
          f"(host, one core: {t_dec / F * 1e3:.1f} ms)")
    print(f"per frame: decode 6 JPEGs {t_dec / F * 1e3:.1f} ms | upload 26 MB {t_up / F * 1e3:.2f} ms | "
          f"device (resample+project+stamp+mosaic, 1 frame/launch) {t_gpu / F * 1e3:.3f} ms | "
          f"download 9.3 MB {t_down / F * 1e3:.2f} ms")
    assert host.shape == (1, 1080, 2880, 3)


if __name__ == "__main__":
    main()
