#!/usr/bin/env python3
"""Ingest alone: the decode pump of ClipFrameSource (file reads -> pinned memory -> device JPEG decode) consumed as fast as
possible, no render, no egress: six-camera frames/s of a planned pass over a 1600x900 JPEG clip.

    python tools/ingest_probe.py [--frames 240] [--batch 16]"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--batch", type=int, default=16)
    args = ap.parse_args()
    import torch
    from cama.dataset import ClipManager
    from cama_amd.synth import DEFAULT_CAMA_CONFIGS
    from tools.loop_timeline import fast_jpeg_clip
    root = tempfile.mkdtemp(prefix="cama_ingest_")
    clip = os.path.join(root, "clip")
    fast_jpeg_clip(clip, args.frames + 1)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)
    src = cm.frame_source()
    ids = list(range(1, args.frames + 1))
    batches = [ids[k:k + args.batch] for k in range(0, len(ids), args.batch)]
    for rep in range(5):
        src.plan(batches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        waits = 0.0
        for b in batches:
            t1 = time.perf_counter()
            raw = src.raw_batch(b)
            waits += time.perf_counter() - t1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"ingest only: {args.frames} frames in {dt * 1e3:.1f} ms = {args.frames / dt:.0f} frames/s "
              f"({6 * args.frames / dt:.0f} images/s); consumer waited {waits * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
