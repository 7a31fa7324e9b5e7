"""SURVEY 8(b) "Threading": the library must be "safe for many processes sharing one GPU" -- nginx runs worker_processes N
(docs/02 - Configuration.md:18), each a process of its own that calls OnEnvStart after the fork (module.c:100-107).  Four
fresh processes take requests at the same time -- JPEG in, resize=224,0, JPEG out, all of it on the device, the entropy
stage included (its workgroups wait for each other: the one kernel where time-slicing between processes could matter) --
and every answer must be the file the oracle writes; no wait between workgroups may run out."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _photo(h, w, seed):
    sys.path.insert(0, os.path.dirname(HERE))
    from ngx_http_imgproc_amd.workloads import photo_like
    return photo_like(h, w, seed)[:, :, ::-1].copy()          # B,G,R


def _run(d, nproc, count):
    go = os.path.join(d, "go")
    for f in os.listdir(d):
        if f.startswith("ready.") or f == "go":
            os.unlink(os.path.join(d, f))
    env = dict(os.environ)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multiproc_worker.py"), d, str(i), str(count), go], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(nproc)]
    t_end = time.time() + 240
    while sum(os.path.exists(os.path.join(d, "ready.%d" % i)) for i in range(nproc)) < nproc:
        assert time.time() < t_end and all(p.poll() is None for p in procs), [p.stderr.read()[-400:] for p in procs if p.poll() is not None]
        time.sleep(0.01)
    open(go, "w").close()
    out = []
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se[-800:]
        out.append(json.loads(so.strip().splitlines()[-1]))
    return out


def test_four_worker_processes_share_the_device():
    sizes = [(480, 640, 1), (720, 1280, 2), (1080, 1920, 3), (600, 800, 4), (1200, 1600, 5)]
    with tempfile.TemporaryDirectory() as d:
        for k, (h, w, seed) in enumerate(sizes):
            rc, blob = orc.jpeg_encode(_photo(h, w, seed), 90)
            assert rc == 0
            rc, frame = orc.jpeg_decode(blob)
            assert rc == 0
            rc, small = orc.resize(frame, "224,0")
            assert rc == 0
            rc, answer = orc.jpeg_encode(small, 86)
            assert rc == 0
            open(os.path.join(d, "f%d.jpg" % k), "wb").write(blob)
            open(os.path.join(d, "f%d.out" % k), "wb").write(answer)
        one = _run(d, 1, 50)
        four = _run(d, 4, 50)
    for r in one + four:
        assert r["mismatches"] == 0, r
        assert r["chain_timeouts"] == 0 and r["refused"] == 0, r
        assert r["device_entropy_files"] >= r["requests"], r
    rate1 = one[0]["requests"] / one[0]["seconds"]
    rate4 = sum(r["requests"] for r in four) / max(r["seconds"] for r in four)
    print("\none process: %.0f requests/s; four processes sharing the device: %.0f requests/s together" % (rate1, rate4))
    assert rate4 > 0.8 * rate1            # sharing must not collapse the device (it should go up: one request at a time leaves it idle)
