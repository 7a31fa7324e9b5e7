#!/usr/bin/env python3
"""Throughput of the 3-channel (BGR, what cvDecodeImage gives for a JPEG) resize paths on resident 1080p frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngx_http_imgproc_amd as imp

torch.cuda.set_device(0)
imp.env_start(0)
n = int(os.environ.get("PROBE_BATCH", "512"))
stream = torch.cuda.Stream()
for c in (3, 4):
    src = torch.randint(0, 256, (n, 1080, 1920, c), dtype=torch.uint8, device="cuda")
    dst = torch.zeros((n, 224, 224, c), dtype=torch.uint8, device="cuda")
    for name, interp in (("cubic", imp.INTER_CUBIC), ("area", imp.INTER_AREA), ("linear", imp.INTER_LINEAR), ("nn", imp.INTER_NN)):
        def step():
            imp.batch_cv_resize(src.data_ptr(), 1080 * 1920 * c, 1920, 1080, 1920 * c, dst.data_ptr(), 224 * 224 * c, 224, 224,
                                224 * c, c, n, interp, stream=stream.cuda_stream)
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
        print("c=%d %-7s %8.3f ms / %d frames  %9.0f img/s" % (c, name, ms, n, n / ms * 1e3), flush=True)
    del src, dst
imp.env_destroy()
