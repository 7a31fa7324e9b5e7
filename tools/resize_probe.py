#!/usr/bin/env python3
"""Throughput of one batched cvResize geometry on resident frames:  tools/resize_probe.py SW SH DW DH C INTERP [BATCH]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngx_http_imgproc_amd as imp

sw, sh, dw, dh, c, interp = [int(v) for v in sys.argv[1:7]]
n = int(sys.argv[7]) if len(sys.argv) > 7 else 256
torch.cuda.set_device(0)
imp.env_start(0)
stream = torch.cuda.Stream()
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
for _ in range(20):
    step()
e1.record(stream)
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
gb = n * (sh * sw + dh * dw) * c / 1e9
print("%dx%d -> %dx%d c=%d interp=%d: %8.3f ms / %d frames  %9.0f img/s  %7.1f GB/s alg (%.3f of 8 TB/s)" %
      (sw, sh, dw, dh, c, interp, ms, n, n / ms * 1e3, gb / ms * 1e3, gb / ms * 1e3 / 8000), flush=True)
imp.env_destroy()
