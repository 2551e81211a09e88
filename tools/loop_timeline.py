#!/usr/bin/env python3
"""Where the verbatim main.py loop spends its time (host wall clock per stage + frames/s), on a 1600x900 JPEG clip.

    CAMA_VIDEO_SINK=null python tools/loop_timeline.py [--frames 240] [--passes 3] [--render-ahead 16]

Stages are timed by wrapping the product's own functions (no change to the loop): waiting for file reads, submitting the
device JPEG decode, waiting for a decoded batch, issuing the render, issuing the egress (BGR -> I420 + download), waiting
for the I420 bytes, writing them to the sink, everything else (the loop's Python: yield_frame, handles, tqdm).
The clip is written quickly: 8 distinct photo-like frames per camera, hard-linked over the clip's timestamps (the decode
work per file is what a real clip's would be; only the page cache sees less variety)."""
import argparse
import collections
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CAMA_VIDEO_SINK", "null")


def fast_jpeg_clip(clip, n_frames, distinct=8, seed=0):
    import json
    import shutil
    from cama_amd.synth import CAMERA_NAMES, make_clip
    make_clip(clip, n_frames=n_frames, seed=seed, n_lines=20, verts_per_line=11, line_len_m=5.0, raster_size=3000,
              image_mode="none", image_size=(900, 1600), with_nuscenes=False, extra_labels=False)
    tmp = clip + "_imgs"
    make_clip(tmp, n_frames=distinct, seed=seed, n_lines=2, verts_per_line=3, line_len_m=1.0, raster_size=64,
              image_mode="jpg_photo", image_size=(900, 1600), with_nuscenes=False, extra_labels=False)
    att = json.load(open(os.path.join(clip, "attribute.json")))
    att_t = json.load(open(os.path.join(tmp, "attribute.json")))
    for name in CAMERA_NAMES:
        os.makedirs(os.path.join(clip, name), exist_ok=True)
        src = [os.path.join(tmp, name, f"{ts}.jpg") for ts in att_t["sync"][name]]
        for k, ts in enumerate(att["sync"][name]):
            dst = os.path.join(clip, name, f"{ts}.jpg")
            try:
                os.link(src[k % distinct], dst)
            except OSError:
                shutil.copy(src[k % distinct], dst)


class Clock:
    """Exclusive wall time per wrapped function, per THREAD (the decode pump runs beside the loop): labels of another
    thread than the loop's get a "[pump]" prefix and do not add up with the loop's."""

    def __init__(self):
        import threading
        self.t = collections.OrderedDict()
        self.n = collections.Counter()
        self.events = []
        self.local = threading.local()
        self.main = threading.get_ident()

    def wrap(self, obj, name, label):
        fn = getattr(obj, name)
        clock = self

        def timed(*a, **k):
            import threading
            stack = clock.local.__dict__.setdefault("stack", [])
            lab = label if threading.get_ident() == clock.main else "[pump] " + label
            t0 = time.perf_counter()
            stack.append(0.0)
            try:
                return fn(*a, **k)
            finally:
                dt = time.perf_counter() - t0
                inner = stack.pop()
                clock.t[lab] = clock.t.get(lab, 0.0) + dt - inner          # exclusive time
                clock.n[lab] += 1
                clock.events.append((t0, t0 + dt, lab))
                if stack:
                    stack[-1] += dt
        setattr(obj, name, timed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--render-ahead", type=int, default=16)
    ap.add_argument("--events", action="store_true", help="print the instrumented pass as a timeline of calls")
    ap.add_argument("--no-egress", action="store_true", help="A/B: no VideoGenerator (no I420 conversion, no download)")
    args = ap.parse_args()
    import torch
    from cama.dataset import ClipManager
    from cama.tools import VideoGenerator
    from cama_amd import egress, frames, jpeg
    from cama_amd.synth import DEFAULT_CAMA_CONFIGS
    root = tempfile.mkdtemp(prefix="cama_loop_")
    clip = os.path.join(root, "clip")
    t = time.perf_counter()
    fast_jpeg_clip(clip, args.frames + 1)
    print(f"clip written in {time.perf_counter() - t:.1f} s")
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS, render_ahead=args.render_ahead), clip)

    def one_pass():
        if args.no_egress:
            n = 0
            for image_idx, instance_map in cm.yield_frame(dataset="cama"):
                cm.render_vectors(cm.project_all_camera(instance_map), image_idx)
                n += 1
            torch.cuda.synchronize()
            return n
        vg = VideoGenerator(os.path.join(root, "out.mp4"), (2880, 1080))
        n = 0
        for image_idx, instance_map in cm.yield_frame(dataset="cama"):
            maps_2d_dict = cm.project_all_camera(instance_map)
            image_dict = cm.render_vectors(maps_2d_dict, image_idx)
            image = vg.concate_image(image_dict)
            vg.add_frame(image)
            n += 1
        vg.close()
        return n

    one_pass()
    one_pass()
    torch.cuda.synchronize()
    for _ in range(args.passes):
        t0 = time.perf_counter()
        n = one_pass()
        dt = time.perf_counter() - t0
        print(f"un-instrumented: {n} frames in {dt * 1e3:.1f} ms = {n / dt:.0f} frames/s")
    ck = Clock()
    ck.wrap(frames.ClipFrameSource, "_collect", "wait for file reads (_collect)")
    ck.wrap(frames.ClipFrameSource, "_submit_batch", "queue a batch's file reads")
    ck.wrap(jpeg.DeviceJpegDecoder, "decode_async", "submit device JPEG decode (parse + upload + launches)")
    ck.wrap(jpeg.DeviceJpegDecoder, "decode", "device JPEG decode, synchronous path")
    ck.wrap(frames.ClipFrameSource, "_pump_step", "pump step (exclusive: waiting for the batch's file reads)")
    ck.wrap(jpeg.DeviceJpegDecoder, "_submit", "decode group submit (pack + upload + launches)")
    ck.wrap(jpeg.PendingDecode, "result", "PendingDecode.result: wait for the GPU decode of a batch")
    ck.wrap(frames.ClipFrameSource, "raw_batch", "raw_batch: wait for a decoded batch + bookkeeping")
    ck.wrap(ClipManager, "_render_batch", "issue render (cama_render_frames / raw overlay)")
    ck.wrap(egress.RenderBatch, "start_egress", "issue egress (BGR->I420 + async download)")
    ck.wrap(egress.RenderBatch, "i420", "wait for I420 bytes (event sync)")
    ck.wrap(VideoGenerator, "add_frame", "add_frame: sink write + glue")
    ck.wrap(ClipManager, "frame_poses", "frame_poses (seek + slerp + inverse, whole pass)")
    ck.wrap(ClipManager, "_render_ahead", "_render_ahead bookkeeping")
    t0 = time.perf_counter()
    n = one_pass()
    total = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"instrumented pass: {n} frames in {total * 1e3:.1f} ms = {n / total:.0f} frames/s")
    acc = 0.0
    for label, sec in sorted(ck.t.items(), key=lambda kv: -kv[1]):
        if not label.startswith("[pump]"):
            acc += sec
        print(f"  {sec * 1e3:8.2f} ms  {100 * sec / total:5.1f} %  x{ck.n[label]:<5d} {label}")
    print(f"  {(total - acc) * 1e3:8.2f} ms  {100 * (total - acc) / total:5.1f} %         loop Python (yield_frame, handles, tqdm, concate_image)")
    if args.events:
        print("timeline of the instrumented pass (ms from its start; begin..end label), calls of >= 0.15 ms:")
        for a, b, lab in sorted(ck.events):
            if b - a >= 0.15e-3 and "bookkeeping" not in lab[:16]:
                print(f"  {(a - t0) * 1e3:7.2f} .. {(b - t0) * 1e3:7.2f}  {lab[:70]}")


if __name__ == "__main__":
    main()
