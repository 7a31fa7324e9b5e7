#!/usr/bin/env python3
"""cfg3 chain kernel next to a device copy of the same read volume, for TCC write-path counters.
--dstep N pads the destination row pitch (default 540*4 = 2160, not a multiple of 128)."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ngx_http_imgproc_amd as imp
ap = argparse.ArgumentParser()
ap.add_argument("--dstep", type=int, default=2160)
ap.add_argument("--frames", type=int, default=512)
ap.add_argument("--time", type=int, default=0, help="time this many launches with events instead of a profiler run")
a = ap.parse_args()
torch.cuda.set_device(0)
imp.env_start(0)
n = a.frames
src = torch.randint(0, 256, (n, 1080, 1920, 4), dtype=torch.uint8, device="cuda")
fstride = (960 * a.dstep + 127) // 128 * 128
dst = torch.zeros((n, fstride), dtype=torch.uint8, device="cuda")
cfg = imp.Config()
stream = torch.cuda.Stream()
torch.cuda.synchronize()


def chain():
    imp.batch_resize_rotate_watermark(src.data_ptr(), 1080 * 1920 * 4, 1920, 1080, 1920 * 4, dst.data_ptr(), fstride,
                                      a.dstep, 960, 540, 90, cfg, 4, n, stream=stream.cuda_stream)


with torch.cuda.stream(stream):
    if a.time:
        for _ in range(5):
            chain()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(a.time):
            chain()
        e1.record(stream)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / a.time
        print("dstep %d frames %d: %.4f ms/launch, %.0f img/s" % (a.dstep, n, ms, n / ms * 1e3))
    else:
        cpy = torch.empty_like(src)
        for _ in range(2):
            cpy.view(torch.int32).copy_(src.view(torch.int32))
            chain()
torch.cuda.synchronize()
imp.env_destroy()
