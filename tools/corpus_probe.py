"""What share of a directory of JPEG files does the device decoder take, what share goes back to cvDecodeImage (bridge.c:545-552)
and why -- and what does that fallback cost the host?  (Round 5, review item 6: know what the fallback costs before widening
the decoder.)  Host only: impgpu_jpeg_classify reads the headers, Pillow (libjpeg-turbo) times the decode of the refused files
on one core.
    python tools/corpus_probe.py DIR [DIR ...]          every *.jpg / *.jpeg below the directories
    python tools/corpus_probe.py --box                  the JPEG files this image ships (python packages' sample data)"""
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = ["taken by the device", "progressive", "arithmetic / lossless / hierarchical", "12-bit", "CMYK / YCCK / other component count",
         "not one interleaved scan", "sampling factors", "other", "damaged header"]
BOX_DIRS = ["/usr/local/lib/python3.10/dist-packages", "/usr/share", "/opt/conda/lib/python3.9/site-packages", "/opt/conda/doc"]


def find(dirs):
    out = []
    for d in dirs:
        for root, _, files in os.walk(d):
            for f in files:
                if f.lower().endswith((".jpg", ".jpeg")):
                    out.append(os.path.join(root, f))
    return sorted(set(out))


def main():
    from PIL import Image
    from ngx_http_imgproc_amd._lib import lib

    dirs = BOX_DIRS if sys.argv[1:] == ["--box"] else sys.argv[1:]
    if not dirs:
        raise SystemExit(__doc__)
    files = find(dirs)
    buckets = [{"files": 0, "bytes": 0, "pixels": 0, "host_ms": 0.0, "names": []} for _ in NAMES]
    for path in files:
        blob = open(path, "rb").read()
        b = lib.impgpu_jpeg_classify(blob, len(blob))
        B = buckets[b]
        B["files"] += 1
        B["bytes"] += len(blob)
        B["names"].append(os.path.basename(path))
        try:
            best = None
            for _ in range(3):                                   # best of three: the first pass pays for page faults and imports
                t0 = time.perf_counter()
                im = Image.open(io.BytesIO(blob))
                im.load()
                dt = 1e3 * (time.perf_counter() - t0)
                best = dt if best is None else min(best, dt)
            B["host_ms"] += best
            B["pixels"] += im.size[0] * im.size[1]
        except Exception:
            pass
    total = max(len(files), 1)
    total_ms = sum(B["host_ms"] for B in buckets) or 1.0
    print("%d files" % len(files))
    for name, B in zip(NAMES, buckets):
        if B["files"]:
            print("  %-40s %4d files = %5.1f %%, %8.1f KB, %7.2f Mpx, host decode %8.2f ms on one core = %5.1f %% of the corpus' host decode time   e.g. %s"
                  % (name, B["files"], 100.0 * B["files"] / total, B["bytes"] / 1e3, B["pixels"] / 1e6, B["host_ms"], 100.0 * B["host_ms"] / total_ms, ", ".join(B["names"][:3])))
    print(json.dumps({n: {k: v for k, v in B.items() if k != "names"} for n, B in zip(NAMES, buckets) if B["files"]}))


if __name__ == "__main__":
    main()
