#!/usr/bin/env python3
"""Turn rocprofv3 output dirs (gpurun_out/prof_trace, prof_fetch, prof_write) into the tracked summaries under profiles/.

    python tools/summarize_prof.py r01

Writes profiles/<round>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim), profiles/<round>_pmc.json
(per-kernel FETCH_SIZE / WRITE_SIZE, raw and corrected) and profiles/traffic_<mode>.json (what bench.py reports as
roofline.traffic).  Correction, per MI355X_MICROARCH.md "HBM": FETCH_SIZE counts 128-byte requests as 64 on gfx950,
so read bytes = FETCH_SIZE * 2; WRITE_SIZE is exact; both are in KiB.  The device copy in tools/pmc_probe.py (known
byte count) is the calibration and is checked here.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"

MODES = {"k_resize_taps<4, 4, 1>": "cubic", "k_resize_area_rows<4, 10>": "area", "k_resize_nn<4>": "nn",
         "k_resize_taps<2, 4, 0>": "linear", "k_resize_2x_dma<8, 2": "lanczos", "k_area2x2_turn": "chain",
         "k_resize_up_cubic4": "upscale", "k_area2x2_c4": "area2x",
         "k_resize_strip<4, 1, 4>": "upscale_x", "k_resize_strip2<2, 0, 4,": "linear_up",
         "k_resize_strip2<8, 2, 4, 1, 0>": "lanczos_up", "k_resize_strip2<8, 2, 4, 0, 1>": "lanczos_up",
         "k_resize_strip2<8, 2, 4, 1, 2>": "lanczos_15", "k_resize_strip2<8, 2, 4, 2, 1>": "lanczos_15",
         "k_blur_mfma_fused<4, false>": "blur_sigma2", "k_blur_mfma_rows<4, false>": "blur_rows", "k_blur_mfma_cols<4, false>": "blur_cols"}
# (round 4: the 2x enlargements and the 1.5x reduction run k_resize_strip2 with different advance patterns, so their
# dispatches no longer share a kernel name; the blur kernels' entries are per launch of ONE 1080p frame, batch 1)
# frames per launch in tools/pmc_probe.py, as a divisor of PROBE_BATCH
BATCH_DIV = {"lanczos": 16, "upscale": 2, "area2x": 4, "upscale_x": 4, "lanczos_up": 4, "linear_up": 4, "lanczos_15": 16,
             "blur_sigma2": 1024, "blur_rows": 1024, "blur_cols": 1024}


def counter_means(kind):
    files = glob.glob(os.path.join(OUT, "prof_%s" % kind, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(list)
    for f in files:
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
        for r in rows:
            name = r["Kernel_Name"]
            agg[name].append(float(r["Counter_Value"]))
    return agg


def main():
    os.makedirs(PROF, exist_ok=True)
    stats = glob.glob(os.path.join(OUT, "prof_trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(PROF, "%s_kernel_stats.csv" % tag))
    fetch, write = counter_means("fetch"), counter_means("write")
    batch = int(os.environ.get("PROBE_BATCH", "1024"))
    src_bytes = batch * 1080 * 1920 * 4
    summary = {"units": "bytes per launch", "batch": batch, "kernels": {}}
    # calibration on the device copy: keep the dispatches that moved the whole batch
    cal_f = [v for v in fetch.get("__amd_rocclr_copyBuffer", []) if v * 1024 > src_bytes / 4]
    cal_w = [v for v in write.get("__amd_rocclr_copyBuffer", []) if v * 1024 > src_bytes / 4]
    if cal_f and cal_w:
        summary["calibration"] = {
            "copy_bytes_each_way": src_bytes,
            "FETCH_SIZE_raw_bytes": sum(cal_f) / len(cal_f) * 1024,
            "WRITE_SIZE_raw_bytes": sum(cal_w) / len(cal_w) * 1024,
            "fetch_scale": src_bytes / (sum(cal_f) / len(cal_f) * 1024),
            "write_scale": src_bytes / (sum(cal_w) / len(cal_w) * 1024),
        }
    for k in sorted(set(fetch) | set(write)):
        if "imp::" not in k:
            continue
        f = sum(fetch[k]) / max(1, len(fetch[k])) * 1024 if k in fetch else None
        w = sum(write[k]) / max(1, len(write[k])) * 1024 if k in write else None
        entry = {"FETCH_SIZE_raw_bytes": f, "WRITE_SIZE_raw_bytes": w,
                 "hbm_read_bytes": None if f is None else f * 2, "hbm_write_bytes": w,
                 "hbm_bytes_per_launch": None if f is None or w is None else f * 2 + w}
        summary["kernels"][k] = entry
        for pat, mode in MODES.items():
            if pat in k and entry["hbm_bytes_per_launch"]:
                with open(os.path.join(PROF, "traffic_%s.json" % mode), "w") as fh:
                    json.dump({"kernel": k, "round": tag, "batch": batch // BATCH_DIV.get(mode, 1),
                               "hbm_bytes_per_launch": round(entry["hbm_bytes_per_launch"]),
                               "read_bytes": round(f * 2), "write_bytes": round(w),
                               "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                                      "tools/pmc_probe.py; KiB -> bytes; FETCH_SIZE doubled (gfx950 counts 128-B "
                                      "requests as 64 B; checked on the same run's device copy)"}, fh, indent=1)
    with open(os.path.join(PROF, "%s_pmc.json" % tag), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary.get("calibration"), indent=1))
    for k, v in summary["kernels"].items():
        print("%-90s %.3f GB/launch" % (k[:90], (v["hbm_bytes_per_launch"] or 0) / 1e9))


if __name__ == "__main__":
    main()
