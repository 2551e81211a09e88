"""Fuzz of the anti-aliased segment extension: random polyline scenes, sizes and radii, HIP against oracle_render_frame_wu.
    python tools/fuzz_wu.py [--seeds 24]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=24)
    args = ap.parse_args()
    import torch
    from cama_amd import runtime
    from oracle import cama_oracle as O
    from tests.test_gpu_kernels import _polyline_scene
    eng = runtime.engine()
    bad = 0
    for seed in range(args.seeds):
        W, H = [(320, 180), (480, 272), (960, 540), (1600, 900), (640, 368)][seed % 5]
        F = 1 + seed % 3
        xyz, col, link, cams, w2c = _polyline_scene(1000 + seed, W, H, F, n_dense=8 + seed % 20, n_sparse=4 + seed % 30, n_single=seed % 4)
        rig = eng.make_rig([c["name"] for c in cams], [c["chassis2camera"] for c in cams], [c["K"] for c in cams], W, H)
        dmap = eng.upload_map(xyz, col | (link.astype(np.uint8) << 1), spatial_sort=False)
        src = torch.randint(0, 256, (F, 6, H, W, 3), dtype=torch.uint8, device="cuda")
        got = eng.render_frames(dmap, rig, w2c, src, segments="wu").cpu().numpy()
        host = src.cpu().numpy()
        for f in range(F):
            flat = O.frame_project_flat(xyz, w2c[f], cams, W, H)
            want = O.frame_render_flat_wu(host[f], flat["vu"], flat["vis"], col, link)
            n = int(np.count_nonzero((got[f] != want).any(axis=2)))
            if n:
                bad += 1
                print(f"seed {seed} {W}x{H} frame {f}: {n} pixels differ")
    print(f"fuzz_wu: {args.seeds} scenes, {bad} frames with differences")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
