#!/usr/bin/env python3
"""Short workload for SQ/TCP counter passes: cubic + area on 256 1080p frames, lanczos on 32 4K frames."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ngx_http_imgproc_amd as imp  # noqa: E402

torch.cuda.set_device(0)
imp.env_start(0)
g = torch.Generator(device="cuda")
g.manual_seed(1)
stream = torch.cuda.Stream()
n1, n2 = 256, 32
src = torch.randint(0, 256, (n1, 1080, 1920, 4), dtype=torch.uint8, device="cuda", generator=g)
dst = torch.zeros((n1, 224, 224, 4), dtype=torch.uint8, device="cuda")
src4k = torch.randint(0, 256, (n2, 2160, 3840, 4), dtype=torch.uint8, device="cuda", generator=g)
dst4k = torch.zeros((n2, 1080, 1920, 4), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
with torch.cuda.stream(stream):
    for _ in range(2):
        for interp in (imp.INTER_CUBIC, imp.INTER_AREA):
            imp.batch_cv_resize(src.data_ptr(), 1080 * 1920 * 4, 1920, 1080, 1920 * 4, dst.data_ptr(), 224 * 224 * 4,
                                224, 224, 224 * 4, 4, n1, interp, stream=stream.cuda_stream)
        imp.batch_cv_resize(src4k.data_ptr(), 2160 * 3840 * 4, 3840, 2160, 3840 * 4, dst4k.data_ptr(), 1080 * 1920 * 4,
                            1920, 1080, 1920 * 4, 4, n2, imp.INTER_LANCZOS4, stream=stream.cuda_stream)
torch.cuda.synchronize()
imp.env_destroy()
