"""Does the 3:5 raw overlay's speed depend on where its mosaic (and raw frames) sit, like the plain overlay's does?
One process, the raw-frame headline scene, the same launch into 10 different mosaic allocations, then from 6 different copies
of the raw frames: mean step time over 30 steps each (whole pipelined step, events on the current stream)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from cama_amd.frames import RawDeviceFrameSource
    args = bench.parse_args(["--raw-frames", "--height", "540", "--width", "960", "--cpu-seconds", "0", "--audition", "0"])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    job = bench.Job(args, [0], dev)

    def timed(n=30):
        for _ in range(5):
            job.step()
        job.eng.join()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            job.step()
        job.eng.join()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n

    keep = []
    res = []
    for k in range(10):
        res.append(round(timed(), 4))
        keep.append(job.out)
        job.out = torch.empty_like(job.out)
    print("ms per step into 10 mosaic allocations:", res)
    sid, cm, frames, clip = job.scenes[0]
    res = []
    for k in range(6):
        res.append(round(timed(), 4))
        keep.append(frames)
        frames = frames.clone()
        cm.set_frame_source(RawDeviceFrameSource(frames, cm.cm_list, fused=True))
    print("ms per step from 6 copies of the raw frames:", res)


if __name__ == "__main__":
    main()
