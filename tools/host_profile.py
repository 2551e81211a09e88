"""Where the host spends its time in one pipelined step (cProfile over N steps of bench.Job.step).
    python tools/host_profile.py [bench.py arguments]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    args = bench.parse_args(sys.argv[1:])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    job = bench.Job(args, list(range(max(1, args.scenes))) if args.scenes else [0], dev)
    for _ in range(10):
        job.step()
    job.eng.join()
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        job.step()
    job.eng.join()
    torch.cuda.synchronize()
    print("plain: %.1f us per step" % ((time.perf_counter() - t0) / n * 1e6))
    # host time inside the library's two per-step entry points
    lib = job.eng.lib
    acc, originals = {}, {}
    for name in ("cama_pipeline_render", "cama_pipeline_stage_poses", "cama_pipeline_issued", "cama_pipeline_completed"):
        orig = originals[name] = getattr(lib, name)

        def timed(*a, _orig=orig, _name=name):
            t = time.perf_counter()
            r = _orig(*a)
            acc[_name] = acc.get(_name, 0.0) + time.perf_counter() - t
            return r
        setattr(lib, name, timed)
    t0 = time.perf_counter()
    for _ in range(n):
        job.step()
    job.eng.join()
    torch.cuda.synchronize()
    print("with the library calls timed: %.1f us per step; inside " % ((time.perf_counter() - t0) / n * 1e6) +
          ", ".join("%s %.1f us" % (k, v / n * 1e6) for k, v in acc.items()))
    for name, orig in originals.items():        # (put the typed ctypes functions back: a fresh lookup would have no argtypes)
        setattr(lib, name, orig)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        job.step()
    pr.disable()
    job.eng.join()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
