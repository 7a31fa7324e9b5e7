"""CPU side of the broker (include/impgpu_broker.h): the client fails loudly and fast when nobody serves, the broker fails
loudly without a device (no CPU path behind it either), the records are one page each."""
import ctypes as C
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built():
    subprocess.check_call([sys.executable, os.path.join(ROOT, "ngx_http_imgproc_amd", "build.py")], stdout=subprocess.DEVNULL)
    from ngx_http_imgproc_amd import broker as B
    return B


def test_client_library_exports_the_declared_symbols(built):
    hdr = open(os.path.join(ROOT, "include", "impgpu_broker.h")).read()
    import re

    names = set(re.findall(r"\b(impgpu_client_[a-z_]+)\s*\(", hdr))
    assert len(names) >= 7
    for n in names:
        assert hasattr(built.clib, n), n
    out = subprocess.run(["ldd", built.CLIENT_LIB_PATH], capture_output=True, text=True).stdout
    assert "amdhip" not in out and "stdc++" not in out          # workers: plain C, no HIP, no C++ runtime


def test_attach_without_a_broker_fails_at_once(built):
    with pytest.raises(RuntimeError) as e:
        built.Client("/impgpu-nobody-%d" % os.getpid())
    assert "no broker segment" in str(e.value)


def test_header_is_c99_and_records_are_pages(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include <impgpu_broker.h>\n#include <stdio.h>\nint main(void){printf("%zu %zu\\n", sizeof(impb_header), sizeof(impb_slot));return 0;}\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert subprocess.check_output([str(exe)], text=True).split() == ["4096", "4096"]


def test_broker_fails_loudly_without_a_device(built):
    import torch

    env = dict(os.environ)
    if torch.cuda.is_available():
        env.update({"HIP_VISIBLE_DEVICES": "-1", "ROCR_VISIBLE_DEVICES": "-1"})
    name = "/impgpu-nodev-%d" % os.getpid()
    p = subprocess.run([built.BROKER_PATH, "--name", name, "--slots", "2", "--slot-mb", "1"], capture_output=True, text=True, timeout=120, env=env)
    try:
        assert p.returncode == 4 and "impgpu_env_start" in p.stderr
        # nobody was ever told the segment is served
        with pytest.raises(RuntimeError) as e:
            built.Client(name)
        assert "no live broker" in str(e.value) or "no broker segment" in str(e.value)
    finally:
        try:
            os.unlink("/dev/shm" + name)
        except OSError:
            pass


def test_broker_rejects_bad_options(built):
    p = subprocess.run([built.BROKER_PATH, "--slots", "100000"], capture_output=True, text=True, timeout=30)
    assert p.returncode == 2
    p = subprocess.run([built.BROKER_PATH, "--name", "no-slash"], capture_output=True, text=True, timeout=30)
    assert p.returncode == 2
