"""Build libcama_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess
from os.path import abspath, dirname, exists, getmtime, join

_HERE = dirname(abspath(__file__))
REPO = dirname(_HERE)
SOURCES = [join(_HERE, "csrc", n) for n in ("cama_hip.hip", "cama_pipeline.hip", "cama_jpeg.hip")]
DEVICE_HEADERS = [join(_HERE, "csrc", n) for n in ("project_kernels.hpp", "remap_device.hpp", "overlay_kernels.hpp",
                                                    "resample_kernels.hpp", "map_kernels.hpp", "jpeg_kernels.hpp", "raw35_kernels.hpp", "egress_kernels.hpp", "cama_common.hpp", "cama_internal.hpp")]
HEADER = join(REPO, "include", "cama_hip.h")
OUT = join(_HERE, "libcama_hip.so")
# -ffp-contract=off: the fp64 FMA chains are written explicitly; nothing else may be fused
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-I" + join(REPO, "include")]


def build(force=False, verbose=False):
    newest = max(getmtime(p) for p in SOURCES + DEVICE_HEADERS + [HEADER])
    if not force and exists(OUT) and getmtime(OUT) >= newest:
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + SOURCES + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
