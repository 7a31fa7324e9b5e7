"""Build recipe for libimpgpu.so (hipcc, gfx950 only) -- used by __graft_entry__.build().

    python ngx_http_imgproc_amd/build.py          # rebuild if any source is newer
    python ngx_http_imgproc_amd/build.py --force
(run it as a script: `python -m ngx_http_imgproc_amd.build` would import the package, which loads the library)

Flags that matter for parity: -ffp-contract=off (no fused multiply-add on host or device:
the reference's float expressions round after every operation) and no fast-math.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libimpgpu.so")
SOURCES = [
    "imp_runtime.hip",
    "imp_resize.hip",
    "imp_geom.hip",
    "imp_pixel.hip",
    "imp_blur.hip",
    "imp_api.cpp",
    "imp_args.cpp",
    "imp_request.cpp",
    "imp_tables.cpp",
]
HEADERS = ["imp_internal.h", os.path.join("..", "..", "include", "impgpu.h")]
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS + [os.path.join("..", "build.py")]:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    srcs = []
    for f in SOURCES:
        path = os.path.join(CSRC, f)
        # .cpp files hold host code only but share headers with the kernels: compile all as HIP
        srcs += ["-x", "hip", path]
    cmd = [_hipcc()] + FLAGS + ["-I", os.path.join(HERE, "..", "include")] + srcs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
    print(LIB)
