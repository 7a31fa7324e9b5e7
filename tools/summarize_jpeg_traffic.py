#!/usr/bin/env python3
"""gpurun_out/prof_jfetch + prof_jwrite (tools/pmc_jpeg_traffic.sh) -> profiles/<round>_jpeg_pmc.json: HBM bytes of every JPEG
kernel per 64-file batch (the dispatches of a kernel are summed over the run and divided by the probe's three batches, so the
figure does not depend on how many launches the decode call makes of a batch), next to the bytes that batch cannot avoid (the files, their coefficient planes once,
the decoded pixels).  FETCH_SIZE is KiB and counts 128-byte requests as 64 on gfx950 (MI355X_MICROARCH.md "HBM"): x 2;
WRITE_SIZE is KiB, exact for 16-byte stores, uncalibrated for narrower ones (the guide says so; ratios between rounds stand).
    python tools/summarize_jpeg_traffic.py r04"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def means(kind):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "prof_%s" % kind, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


fetch, write = means("jfetch"), means("jwrite")
import bench  # noqa: E402

files = bench.jpeg_pool(64)
file_bytes = sum(len(b) for _, _, b in files)
planes = sum(((w + 15) // 16) * ((h + 15) // 16) * 6 * 128 for w, h, _ in files)
pixels = sum(w * h * 3 for w, h, _ in files)
BATCHES = 3                                      # tools/jpeg_pmc_probe.py decodes, resizes and encodes its 64 files three times
out = {"round": tag, "files_per_batch": 64, "units": "bytes per 64-file batch (all dispatches of the run / %d batches)" % BATCHES,
       "algorithmic": {"file_bytes": file_bytes, "coefficient_planes": planes, "decoded_pixels": pixels,
                       "note": "entropy stage: files in + planes out; k_jpeg_pixels: planes in + pixels out"},
       "kernels": {}}
tot_r = tot_w = 0
for k in sorted(set(fetch) | set(write)):
    if "jpeg" not in k:
        continue
    import re
    m = re.search(r"(k_jpeg_\w+(?:<[^>]*>)?)", k)
    short = m.group(1) if m else k[:40]
    r = sum(fetch[k]) / BATCHES * 1024 * 2 if k in fetch else None
    w = sum(write[k]) / BATCHES * 1024 if k in write else None
    out["kernels"][short] = {"launches": len(fetch.get(k, write.get(k, []))), "hbm_read_bytes": None if r is None else round(r),
                             "hbm_write_bytes": None if w is None else round(w)}
    if "enc" not in short:
        tot_r += r or 0
        tot_w += w or 0
out["decode_total"] = {"hbm_read_bytes": round(tot_r), "hbm_write_bytes": round(tot_w),
                       "over_algorithmic": round((tot_r + tot_w) / (file_bytes + 2 * planes + pixels), 2),
                       "algorithmic_bytes": file_bytes + 2 * planes + pixels}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", "%s_jpeg_pmc.json" % tag), "w") as fh:
    json.dump(out, fh, indent=1)
with open(os.path.join(ROOT, "profiles", "traffic_jpeg.json"), "w") as fh:
    json.dump({"round": tag, "files_per_batch": 64, "hbm_bytes_per_launch": round(tot_r + tot_w), "read_bytes": round(tot_r), "write_bytes": round(tot_w),
               "kernels": "k_jpeg_walks + k_jpeg_mend + k_jpeg_select + k_jpeg_write + k_jpeg_dcfix + k_jpeg_pixels over one batch of 64 files",
               "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/jpeg_pmc_probe.py; KiB -> bytes; FETCH_SIZE doubled"}, fh, indent=1)
for k, v in out["kernels"].items():
    print("%-28s read %8.1f MB  write %8.1f MB" % (k, (v["hbm_read_bytes"] or 0) / 1e6, (v["hbm_write_bytes"] or 0) / 1e6))
print("decode: %.1f MB read + %.1f MB written per 64 files = %.2f x (files + planes twice + pixels = %.1f MB)" %
      (tot_r / 1e6, tot_w / 1e6, out["decode_total"]["over_algorithmic"], out["decode_total"]["algorithmic_bytes"] / 1e6))
