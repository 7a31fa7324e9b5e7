"""One of N worker processes sharing a GPU (tests/test_gpu_multiproc.py): impgpu_env_start like an nginx worker after fork
(OnEnvStart, bridge.c:10-16), then `count` requests -- JPEG file in, resize=224,0, JPEG file out -- each answer compared with
the file the parent prepared with the oracle.  Prints one JSON line.
    python tests/multiproc_worker.py <dir> <worker index> <count> <go file>"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngx_http_imgproc_amd as imp  # noqa: E402


def main():
    d, me, count, go = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    names = sorted(n[:-4] for n in os.listdir(d) if n.endswith(".jpg"))
    files = [open(os.path.join(d, n + ".jpg"), "rb").read() for n in names]
    want = [open(os.path.join(d, n + ".out"), "rb").read() for n in names]
    os.environ["IMPGPU_JPEG_HUFF"] = "device"            # every file through the device's entropy stage, whatever its size
    assert imp.env_start(0) == 0
    rc, im = imp.Image.decode_jpeg(files[0])             # the first call pays for pools and staging
    assert rc == 0
    im.release()
    open(os.path.join(d, "ready.%d" % me), "w").close()
    while not os.path.exists(go):                        # all workers start their requests together
        time.sleep(0.002)
    bad = 0
    t0 = time.perf_counter()
    for i in range(count):
        k = (me + i) % len(files)
        rc, im = imp.Image.decode_jpeg(files[k])
        if rc == 0:
            rc = im.resize("224,0")
        answer = b""
        if rc == 0:
            rc, answer = im.encode_jpeg(86)
        if im is not None:
            im.release()
        if rc != 0 or answer != want[k]:
            bad += 1
    dt = time.perf_counter() - t0
    cnt = (C.c_ulonglong * 4)()
    imp.lib.impgpu_jpeg_counters(cnt, 4)
    imp.env_destroy()
    print(json.dumps({"worker": me, "requests": count, "seconds": dt, "mismatches": bad, "device_entropy_files": cnt[0], "refused": cnt[1],
                      "chain_timeouts": cnt[2], "kept_on_host": cnt[3]}))


if __name__ == "__main__":
    main()
