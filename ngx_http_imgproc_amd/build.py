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
    "imp_jpeg.hip",
    "imp_jpeg_enc.hip",
    "imp_png.hip",
    "imp_api.cpp",
    "imp_args.cpp",
    "imp_request.cpp",
    "imp_tables.cpp",
    "imp_jpeg.cpp",
    "imp_jpeg_api.cpp",
    "imp_png.cpp",
    "imp_inflate.cpp",
]
HEADERS = ["imp_internal.h", "imp_jpeg.h", "imp_jpeg_core.h", "imp_jpeg_std.h", "imp_png.h", "imp_inflate.h", os.path.join("..", "..", "include", "impgpu.h")]
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


def _extra_flags():
    return os.environ.get("IMPGPU_EXTRA_FLAGS", "").split()      # A/B builds: -DNAME=value


def _variant():
    """Objects and library of an A/B build (IMPGPU_EXTRA_FLAGS) live apart from the default ones: same-named files built
    with other flags would otherwise look up to date to the next plain build."""
    import hashlib

    extra = _extra_flags()
    return ("_" + hashlib.sha1(" ".join(extra).encode()).hexdigest()[:10]) if extra else ""


OBJ = os.path.join(HERE, "_obj" + _variant())
if _variant():
    LIB = os.environ.get("IMPGPU_LIB") or os.path.join(HERE, "libimpgpu%s.so" % _variant())


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _common_deps():
    return [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]


def needs_build():
    return _stale(LIB, [os.path.join(CSRC, f) for f in SOURCES] + _common_deps())


def build_library(force=False, verbose=False):
    """One object per source (compiled in parallel, only when stale), then one link."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(OBJ, exist_ok=True)
    compile_flags = [f for f in FLAGS if f != "-shared"] + ["-c", "-I", os.path.join(HERE, "..", "include")]
    compile_flags += _extra_flags()               # an A/B build writes libimpgpu_<hash>.so (or $IMPGPU_LIB); load it with IMPGPU_LIB
    jobs = []
    for f in SOURCES:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f + ".o")
        if force or _stale(obj, [src] + _common_deps()):
            # .cpp files hold host code only but share headers with the kernels: compile all as HIP
            jobs.append([_hipcc()] + compile_flags + ["-x", "hip", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as pool:
        list(pool.map(run, jobs))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(OBJ, f + ".o") for f in SOURCES] + ["-lz", "-o", LIB]   # zlib: the host inflate of imp_png.hip
    run(link)
    return LIB


BROKER = os.path.join(HERE, "impgpu_broker")
CLIENT_LIB = os.path.join(HERE, "libimpgpu_client.so")
INCLUDE = os.path.join(HERE, "..", "include")
CLIENT_SRC = os.path.join(HERE, "..", "glue", "imp_gpu_client.c")


def build_broker(force=False, verbose=False):
    """The one-process-per-GPU broker (host C++ over the C ABI, links libimpgpu.so next to it) and the workers' side of its
    protocol (plain C, no HIP) as a shared object for the tests -- in an nginx build the same file is compiled into the module."""
    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    hdrs = [os.path.join(INCLUDE, "impgpu_broker.h"), os.path.join(INCLUDE, "impgpu.h")]
    if force or _stale(CLIENT_LIB, [CLIENT_SRC] + hdrs):
        run(["gcc", "-std=gnu99", "-O2", "-Wall", "-Wextra", "-fPIC", "-shared", "-I", INCLUDE, CLIENT_SRC, "-o", CLIENT_LIB, "-lrt"])
    src = os.path.join(CSRC, "imp_broker.cpp")
    if force or _stale(BROKER, [src, LIB] + hdrs):
        run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-I", INCLUDE, src, "-o", BROKER, "-L", HERE, "-limpgpu", "-pthread", "-lrt",
             "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link,/opt/rocm/lib"])
    return BROKER


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
    build_broker(force="--force" in sys.argv, verbose=True)
    print(LIB)
