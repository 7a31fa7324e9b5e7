"""glue/ (SURVEY 8f N3): the nginx-side binding as source.  It cannot be BUILT here (nginx / OpenCV / FreeImage headers and
libraries are absent); what is checked is that glue/apply_glue.sh applies to the reference revision it names, leaves RunJob
calling the glue instead of the five CPU loops (and advancedio.c's LoadGIF / LoadSingle / SaveSingle handing frames to and
from the device), and that the result COMPILES: `gcc -fsyntax-only -std=gnu99` over the patched bridge.c, advancedio.c and
glue/imp_gpu_bridge.c against tests/c/decls/ -- declaration-only stand-ins for those headers, test
scaffolding for this one compile check (no oracle, no _ref, nothing is linked or run).  The reference only exists in the
build container: skipped elsewhere."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
def test_apply_glue_rewrites_the_operator_segment(tmp_path):
    work = tmp_path / "module"
    work.mkdir()
    for name in os.listdir(REF):
        if name.endswith((".c", ".h")) or name == "config":
            shutil.copy(os.path.join(REF, name), work / name)
    subprocess.check_call([os.path.join(ROOT, "glue", "apply_glue.sh"), str(work)])
    bridge = (work / "bridge.c").read_text()
    run_job = bridge[bridge.index("RunJob("):]
    for call in ("ImpGpuDecode(blob, size, &album, &gpu", "ImpGpuOperators(&album, &gpu", "ImpGpuInfo(&gpu", "ImpGpuASCII(&gpu", "ImpGpuDownload(&gpu", "ImpGpuEncodeJpeg(&gpu, basicCoderopt[1]", "ImpGpuRelease(&gpu"):
        assert run_job.count(call) == 1, call
    for gone in ("Crop(&image", "Resize(&image", "Filter(&image", "Watermark(image", "BlendWithPaper(image", "CV_GRAY2BGR"):
        assert gone not in run_job, gone
    assert "ImpGpuEnvStart(IMP_GPU_WORKER_INDEX);" in bridge and "ImpGpuEnvDestroy();" in bridge
    assert run_job.index("ImpGpuAlbum gpu = { NULL };") < run_job.index("goto finalize")     # declared before every jump to the release
    assert "cvDecodeImage(&rawencoded, -1)" in run_job                                              # the host decoder stays as the fallback
    assert run_job.index("ImpGpuEncodeJpeg(") < run_job.index("cvEncodeImage(")                       # the host encoder stays for PNG
    assert run_job.count("{") == run_job.count("}")                      # the edit kept the function balanced
    assert "WatermarkDevice;" in (work / "required.h").read_text()
    assert "void*  Device;" in (work / "required.h").read_text()
    assert run_job.index("album.Error = -gpuDecoded;") < run_job.index("cvDecodeImage(&rawencoded, -1)")      # a lost device fails the request, it is not decoded on the host
    assert run_job.count("gpu.Handle = album.Device;") == 1 and run_job.count("album.Device = gpu.Handle;") == 1
    # the FreeImage side (advancedio.c): its three pixel loops are gone, the frames never become IplImages on the way
    adv = (work / "advancedio.c").read_text()
    load_gif = adv[adv.index("static void LoadGIF("):adv.index("static void LoadSingle(")]
    load_single = adv[adv.index("static void LoadSingle("):adv.index("Album FiLoadFrames(")]
    save_single = adv[adv.index("static void SaveSingle("):adv.index("Memory FiSaveFrames(")]
    assert load_gif.count("ImpGpuGifPage(&gif") == 1 and load_gif.count("ImpGpuGifCompose(&gif, isdestructive, page, result)") == 1
    assert "cvSetComponent" not in load_gif and "cvCreateImage" not in load_gif and "master" not in load_gif
    assert "ImpGpuLoadSingle(result, pool, FreeImage_GetBits(fullcolor)" in load_single and "cvSetComponent" not in load_single
    assert "ImpGpuFetchFi(source->Device" in save_single and "IplToFI32(image)" in save_single       # (the host path stays for frames that were downloaded)
    for part in (load_gif, load_single, save_single):
        assert part.count("{") == part.count("}")
    assert "glue/imp_gpu_bridge.c" in (work / "config").read_text() and "-limpgpu" in (work / "config").read_text()
    assert (work / "glue" / "imp_gpu_bridge.c").exists()
    # the patched module compiles: every call matches its prototype, nothing of impgpu.h clashes with required.h, and both
    # spellings of the worker index exist (ngx_worker came with nginx 1.9.1)
    decls = os.path.join(ROOT, "tests", "c", "decls")
    for extra in ([], ["-Dnginx_version=1009005"], ["-Dnginx_version=1004006"]):
        for src in ("bridge.c", "advancedio.c", os.path.join("glue", "imp_gpu_bridge.c"), os.path.join("glue", "imp_gpu_client.c")):
            p = subprocess.run(["gcc", "-fsyntax-only", "-std=gnu99", "-Wall", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
                                "-Werror=int-conversion", "-I", decls, "-I", os.path.join(ROOT, "include"), "-I", str(work)] + extra + [src],
                               cwd=str(work), capture_output=True, text=True)
            assert p.returncode == 0, p.stderr[-3000:]
            assert "warning" not in p.stderr, p.stderr[-3000:]
    # a second application must refuse (the hashes no longer match) instead of editing by stale line numbers
    assert subprocess.call([os.path.join(ROOT, "glue", "apply_glue.sh"), str(work)], stderr=subprocess.DEVNULL) != 0


def test_glue_uses_only_declared_abi_symbols():
    import re

    src = open(os.path.join(ROOT, "glue", "imp_gpu_bridge.c")).read()
    header = open(os.path.join(ROOT, "include", "impgpu.h")).read()
    header += open(os.path.join(ROOT, "include", "impgpu_broker.h")).read()
    used = set(re.findall(r"\b(impgpu_[a-z0-9_]+)\s*\(", src))
    declared = set(re.findall(r"\b(impgpu_[a-z0-9_]+)\s*\(", header))
    assert used and used <= declared, used - declared
