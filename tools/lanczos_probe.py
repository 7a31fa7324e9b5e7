#!/usr/bin/env python3
"""cfg4 kernel alone for counter passes: LANCZOS4 3840x2160 -> 1920x1080 on 32 resident frames, 3 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngx_http_imgproc_amd as imp
torch.cuda.set_device(0)
imp.env_start(0)
n = int(os.environ.get("PROBE_FRAMES", "32"))
interp = {"lanczos": imp.INTER_LANCZOS4, "cubic": imp.INTER_CUBIC}[os.environ.get("PROBE_MODE", "lanczos")]
src = torch.randint(0, 256, (n, 2160, 3840, 4), dtype=torch.uint8, device="cuda")
dst = torch.zeros((n, 1080, 1920, 4), dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.stream(stream):
    for _ in range(3):
        imp.batch_cv_resize(src.data_ptr(), 2160 * 3840 * 4, 3840, 2160, 3840 * 4, dst.data_ptr(), 1080 * 1920 * 4,
                            1920, 1080, 1920 * 4, 4, n, interp, stream=stream.cuda_stream)
torch.cuda.synchronize()
imp.env_destroy()
