#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch / LDS / occupancy of every gfx950 kernel in the library (hipcc -S, no GPU needed).
    python tools/kernel_resources.py > profiles/r01_kernel_resources.txt"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ngx_http_imgproc_amd", "csrc")
rows = []
for f in sorted(os.listdir(CSRC)):
    if not f.endswith(".hip"):
        continue
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-fno-fast-math", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", tmp.name,
                               os.path.join(CSRC, f)], stderr=subprocess.DEVNULL)
        s = open(tmp.name).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel(.*?)(?=\.amdhsa_kernel|\Z)", s, re.S):
        name, desc, tail = m.group(1), m.group(2), m.group(3)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void imp::", "")
        g = lambda pat, txt: (re.search(pat, txt) or [None, "?"])[1]
        rows.append((f, dem, g(r"\.amdhsa_next_free_vgpr (\d+)", desc), g(r"\.amdhsa_next_free_sgpr (\d+)", desc),
                     g(r"; ScratchSize: (\d+)", tail), g(r"; LDSByteSize: (\d+)", tail), g(r"; Occupancy: (\d+)", tail)))
print("%-16s %-58s %5s %5s %7s %7s %4s" % ("file", "kernel", "vgpr", "sgpr", "scratch", "lds", "occ"))
for r in rows:
    print("%-16s %-58s %5s %5s %7s %7s %4s" % r)
