"""glue/ (SURVEY 8f N3): the nginx-side binding as source.  It cannot be compiled here (nginx / OpenCV / FreeImage headers
are absent), so what is checked is that glue/apply_glue.sh applies to the reference revision it names and leaves RunJob
calling the glue instead of the five CPU loops.  The reference only exists in the build container: skipped elsewhere."""
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
    for call in ("ImpGpuOperators(&album, &gpu", "ImpGpuInfo(&gpu", "ImpGpuASCII(&gpu", "ImpGpuDownload(&gpu", "ImpGpuRelease(&gpu"):
        assert run_job.count(call) == 1, call
    for gone in ("Crop(&image", "Resize(&image", "Filter(&image", "Watermark(image", "BlendWithPaper(image", "CV_GRAY2BGR"):
        assert gone not in run_job, gone
    assert "ImpGpuEnvStart((int)ngx_worker);" in bridge and "ImpGpuEnvDestroy();" in bridge
    assert run_job.count("{") == run_job.count("}")                      # the edit kept the function balanced
    assert "WatermarkDevice;" in (work / "required.h").read_text()
    assert "glue/imp_gpu_bridge.c" in (work / "config").read_text() and "-limpgpu" in (work / "config").read_text()
    assert (work / "glue" / "imp_gpu_bridge.c").exists()
    # a second application must refuse (the hashes no longer match) instead of editing by stale line numbers
    assert subprocess.call([os.path.join(ROOT, "glue", "apply_glue.sh"), str(work)], stderr=subprocess.DEVNULL) != 0


def test_glue_uses_only_declared_abi_symbols():
    import re

    src = open(os.path.join(ROOT, "glue", "imp_gpu_bridge.c")).read()
    header = open(os.path.join(ROOT, "include", "impgpu.h")).read()
    used = set(re.findall(r"\b(impgpu_[a-z0-9_]+)\s*\(", src))
    declared = set(re.findall(r"\b(impgpu_[a-z0-9_]+)\s*\(", header))
    assert used and used <= declared, used - declared
