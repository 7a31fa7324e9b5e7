#!/usr/bin/env python3
"""Does the fused 2x2 box + rotate kernel run faster when the rotated destination's row pitch is a multiple of 128 bytes?
(1080p: halved height 540 -> pitch 2160 B, every 256-byte run of a tile straddles line boundaries; 1024 rows: pitch 2048 B.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngx_http_imgproc_amd as imp

torch.cuda.set_device(0)
imp.env_start(0)
stream = torch.cuda.Stream()
cfg = imp.Config()
n = 512
for sh in (1080, 1024, 1152, 1088):
    sw = 1920
    rw, rh = sw // 2, sh // 2
    src = torch.randint(0, 256, (n, sh, sw, 4), dtype=torch.uint8, device="cuda")
    dst = torch.zeros((n, rw, rh, 4), dtype=torch.uint8, device="cuda")

    def step():
        imp.batch_resize_rotate_watermark(src.data_ptr(), sh * sw * 4, sw, sh, sw * 4, dst.data_ptr(), rw * rh * 4, rh * 4,
                                          rw, rh, 90, cfg, 4, n, stream=stream.cuda_stream)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(20):
        step()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gb = n * (sh * sw * 4 + rw * rh * 4) / 1e9
    print("src %dx%d -> rotated %dx%d (pitch %d B, %s): %.3f ms  %.1f GB/s alg  frac %.3f" %
          (sw, sh, rh, rw, rh * 4, "128-aligned" if (rh * 4) % 128 == 0 else "unaligned", ms, gb / ms * 1e3, gb / ms * 1e3 / 8000), flush=True)
    del src, dst
imp.env_destroy()
