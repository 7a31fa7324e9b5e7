#!/usr/bin/env python3
"""Exact-2x resizes on 3-channel vs 4-channel resident frames: AREA 1080p->960x540, CUBIC and LANCZOS4 4K->1080p."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngx_http_imgproc_amd as imp
torch.cuda.set_device(0)
imp.env_start(0)
stream = torch.cuda.Stream()
cases = [("area 1080p->540p", 1920, 1080, 256, imp.INTER_AREA), ("cubic 4K->1080p", 3840, 2160, 64, imp.INTER_CUBIC),
         ("lanczos 4K->1080p", 3840, 2160, 64, imp.INTER_LANCZOS4), ("linear 4K->1080p", 3840, 2160, 64, imp.INTER_LINEAR)]
for c in (3, 4):
    for name, sw, sh, n, interp in cases:
        dw, dh = sw // 2, sh // 2
        src = torch.randint(0, 256, (n, sh, sw, c), dtype=torch.uint8, device="cuda")
        dst = torch.zeros((n, dh, dw, c), dtype=torch.uint8, device="cuda")
        def step():
            imp.batch_cv_resize(src.data_ptr(), sh * sw * c, sw, sh, sw * c, dst.data_ptr(), dh * dw * c, dw, dh, dw * c, c, n, interp,
                                stream=stream.cuda_stream)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            step()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gb = n * (sw * sh + dw * dh) * c / 1e9
        print("c=%d %-18s %8.3f ms / %3d frames  %8.0f img/s  %6.0f GB/s alg" % (c, name, ms, n, n / ms * 1e3, gb / ms * 1e3), flush=True)
        del src, dst
imp.env_destroy()
