#!/usr/bin/env python3
"""Workload for the rocprofv3 passes: a known-size device copy (calibration) followed by the bench kernels.

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/pmc_probe.py
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d OUT -- python3 tools/pmc_probe.py

The copy moves exactly `batch * 1080 * 1920 * 4` bytes each way with 16-byte lanes, which pins the
FETCH_SIZE / WRITE_SIZE scale on this device (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 on gfx950 for
wide coalesced loads) before the resize kernels' counters are read.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import ngx_http_imgproc_amd as imp  # noqa: E402

batch = int(os.environ.get("PROBE_BATCH", "1024"))
reps = int(os.environ.get("PROBE_REPS", "3"))
torch.cuda.set_device(0)
imp.env_start(0)
g = torch.Generator(device="cuda")
g.manual_seed(0x1A4D0001)
src = torch.randint(0, 256, (batch, 1080, 1920, 4), dtype=torch.uint8, device="cuda", generator=g)
cpy = torch.empty_like(src)
dst = torch.zeros((batch, 224, 224, 4), dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
stream.wait_stream(torch.cuda.current_stream())
torch.cuda.synchronize()
with torch.cuda.stream(stream):
    for _ in range(reps):
        cpy.view(torch.int32).copy_(src.view(torch.int32))          # calibration: known bytes
    for interp in (imp.INTER_CUBIC, imp.INTER_AREA, imp.INTER_NN, imp.INTER_LINEAR):
        for _ in range(reps):
            imp.batch_cv_resize(src.data_ptr(), 1080 * 1920 * 4, 1920, 1080, 1920 * 4, dst.data_ptr(), 224 * 224 * 4,
                                224, 224, 224 * 4, 4, batch, interp, stream=stream.cuda_stream)
    # cfg3 chain (fused 2x2 box + rotate, then watermark) and cfg4 Lanczos on smaller batches
    ov = torch.randint(0, 256, (64, 256, 4), dtype=torch.uint8).numpy()
    cfg = imp.Config()
    cfg.prepare_watermark(ov, "r", "b", 16, 16, 60)
    dst3 = torch.zeros((batch, 960, 540, 4), dtype=torch.uint8, device="cuda")
    for _ in range(reps):
        imp.batch_resize_rotate_watermark(src.data_ptr(), 1080 * 1920 * 4, 1920, 1080, 1920 * 4, dst3.data_ptr(), 960 * 540 * 4,
                                          540 * 4, 960, 540, 90, cfg, 4, batch, stream=stream.cuda_stream)
    n4 = max(1, batch // 16)
    src4 = src.view(-1)[: n4 * 2160 * 3840 * 4]
    dst4 = torch.zeros((n4, 1080, 1920, 4), dtype=torch.uint8, device="cuda")
    for _ in range(reps):
        imp.batch_cv_resize(src4.data_ptr(), 2160 * 3840 * 4, 3840, 2160, 3840 * 4, dst4.data_ptr(), 1080 * 1920 * 4,
                            1920, 1080, 1920 * 4, 4, n4, imp.INTER_LANCZOS4, stream=stream.cuda_stream)
    # round 2: the CUBIC enlargement (bridge.c:190's only CUBIC dispatch) on batch/2 frames, the streaming 2x2 box on batch/4
    nu = max(1, batch // 2)
    srcu = src.view(-1)[: nu * 270 * 480 * 4]
    dstu = torch.zeros((nu, 1080, 1920, 4), dtype=torch.uint8, device="cuda")
    for _ in range(reps):
        imp.batch_cv_resize(srcu.data_ptr(), 270 * 480 * 4, 480, 270, 480 * 4, dstu.data_ptr(), 1080 * 1920 * 4,
                            1920, 1080, 1920 * 4, 4, nu, imp.INTER_CUBIC, stream=stream.cuda_stream)
    na = max(1, batch // 4)
    dsta = torch.zeros((na, 540, 960, 4), dtype=torch.uint8, device="cuda")
    for _ in range(reps):
        imp.batch_cv_resize(src.data_ptr(), 1080 * 1920 * 4, 1920, 1080, 1920 * 4, dsta.data_ptr(), 540 * 960 * 4,
                            960, 540, 960 * 4, 4, na, imp.INTER_AREA, stream=stream.cuda_stream)
    # round 3 / 4: the rolling-strip kernel's four bench modes (bench.py WORKLOADS), batch / 4 (batch / 16 for the 2880-wide one)
    for name, (sw, sh, dw, dh, interp, div) in {"upscale_x": (1080, 1920, 1920, 1080, imp.INTER_CUBIC, 4), "lanczos_up": (960, 540, 1920, 1080, imp.INTER_LANCZOS4, 4),
                                                "linear_up": (960, 540, 1920, 1080, imp.INTER_LINEAR, 4), "lanczos_15": (2880, 1620, 1920, 1080, imp.INTER_LANCZOS4, 16)}.items():
        ns = max(1, batch // div)
        ss = src.view(-1)[: ns * sw * sh * 4]
        dd = torch.zeros((ns, dh, dw, 4), dtype=torch.uint8, device="cuda")
        for _ in range(reps):
            imp.batch_cv_resize(ss.data_ptr(), sw * sh * 4, sw, sh, sw * 4, dd.data_ptr(), dw * dh * 4, dw, dh, dw * 4, 4, ns, interp, stream=stream.cuda_stream)
        del dd
torch.cuda.synchronize()
# round 4: filter-blur on the matrix unit, one 1080p BGRA frame per call (sigma 2: one launch; 8 and 16: rows + columns)
frames = [imp.Image(src[i].cpu().numpy()) for i in range(4)]
for sigma in ("2", "8", "16"):
    for im in frames:
        for _ in range(reps):
            assert im.filter("blur=" + sigma, 1) == 0
imp.sync()
for im in frames:
    im.release()
print("probe done: batch", batch, "reps", reps)
imp.env_destroy()
