#!/usr/bin/env python3
"""What the PCIe link gives the main.py loop's egress: device->host copies of a pass's mosaics (bgr24: 9.33 MB per frame at the
reference's default 960x540 output; I420: half of that) into pinned memory, alone and while the pass's compressed frames
(~0.3 MB per 1600x900 JPEG x 6 cameras) are uploaded at the same time.  The loop cannot run faster than bytes / rate.

    python tools/pcie_duplex_probe.py [--frames 240]"""
import argparse
import time

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=240)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = 16                                                     # render-ahead batch (frames per download)
    up_bytes = a.frames * 6 * 300_000
    for name, per_frame in (("bgr24", 1080 * 2880 * 3), ("i420", 1080 * 2880 * 3 // 2)):
        d = torch.empty((B, per_frame), dtype=torch.uint8, device=dev)
        h = [torch.empty((B, per_frame), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        hu = torch.empty(up_bytes // (a.frames // B), dtype=torch.uint8, pin_memory=True)
        du = torch.empty_like(hu, device=dev)
        s_down, s_up = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        n_batches = a.frames // B
        for duplex in (False, True):
            best = 1e9
            for rep in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(n_batches):
                    with torch.cuda.stream(s_down):
                        h[k & 1].copy_(d, non_blocking=True)
                    if duplex:
                        with torch.cuda.stream(s_up):
                            du.copy_(hu, non_blocking=True)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            down = n_batches * B * per_frame
            print(f"{name:6s} {'with uploads' if duplex else 'alone':12s}: {down / 1e9:.2f} GB down"
                  f"{' + %.2f GB up' % (up_bytes / 1e9) if duplex else ''} in {best * 1e3:.1f} ms = {down / best / 1e9:.1f} GB/s down"
                  f" -> at most {a.frames / best:.0f} frames/s for the {a.frames}-frame pass")


if __name__ == "__main__":
    main()
