#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline: images/sec for 1920x1080 -> 224x224 bicubic resize.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode cubic|area|chain|lanczos]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: ONE launch of the resize kernel over
1024 device-resident BGRA frames (BASELINE configs[1]); inputs are in HBM before the timed
region starts.  Multi-GPU: frames are independent, so every rank resizes its own batch with
no data-path collective ("weak" scaling); torch.distributed is used only for the barrier
and the max-over-ranks of the elapsed time.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch
(3 411 968 B per frame for CUBIC) / average launch duration from HIP events
recorded on the launch stream.  `cpu_baseline` times the CPU oracle (oracle/, a port of the
OpenCV 2.4.9 path; the reference itself cannot be built here) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    # name: (src w, src h, dst w, dst h, batch, interpolation id, algorithmic bytes per frame, label)
    "cubic": (1920, 1080, 224, 224, 1024, 2, 896 * 896 * 4 + 224 * 224 * 4,
              "batch 1024 of 1920x1080 BGRA resize->224x224 INTER_CUBIC"),
    "area": (1920, 1080, 224, 224, 1024, 3, 1920 * 1080 * 4 + 224 * 224 * 4,
             "batch 1024 of 1920x1080 BGRA resize->224x224 INTER_AREA (what the reference's Resize() dispatches)"),
    "lanczos": (3840, 2160, 1920, 1080, 64, 4, 3840 * 2160 * 4 + 1920 * 1080 * 4,
                "batch 64 of 3840x2160 BGRA resize->1920x1080 INTER_LANCZOS4"),
    # the reference's only CUBIC dispatch is an enlargement (bridge.c:190): 480x270 -> 1920x1080 `up`; output-dominated
    "upscale": (480, 270, 1920, 1080, 512, 2, 480 * 270 * 4 + 1920 * 1080 * 4,
                "batch 512 of 480x270 BGRA resize->1920x1080 INTER_CUBIC (enlargement: what Resize() sends to CUBIC)"),
    # round 3: the rolling-strip kernel that replaced k_resize_tiled.  CUBIC with one axis growing and the other shrinking
    # (Resize() asks for CUBIC as soon as ONE axis grows, bridge.c:190: a portrait frame made landscape), and the two
    # modes the reference never dispatches but north_star names, enlarging and at a scale between 1 and 2
    "upscale_x": (1080, 1920, 1920, 1080, 256, 2, 1080 * 1920 * 4 + 1920 * 1080 * 4,
                  "batch 256 of 1080x1920 BGRA resize->1920x1080 INTER_CUBIC (x grows 1.78x, y shrinks 1.78x)"),
    "lanczos_up": (960, 540, 1920, 1080, 256, 4, 960 * 540 * 4 + 1920 * 1080 * 4,
                   "batch 256 of 960x540 BGRA resize->1920x1080 INTER_LANCZOS4 (2x enlargement)"),
    "linear_up": (960, 540, 1920, 1080, 256, 1, 960 * 540 * 4 + 1920 * 1080 * 4,
                  "batch 256 of 960x540 BGRA resize->1920x1080 INTER_LINEAR (2x enlargement)"),
    "lanczos_15": (2880, 1620, 1920, 1080, 64, 4, 2880 * 1620 * 4 + 1920 * 1080 * 4,
                   "batch 64 of 2880x1620 BGRA resize->1920x1080 INTER_LANCZOS4 (scale 1.5)"),
    # the simplest shrink: exact 2x2 box (cfg3's resize=960,540 on its own)
    "area2x": (1920, 1080, 960, 540, 256, 3, 1920 * 1080 * 4 + 960 * 540 * 4,
               "batch 256 of 1920x1080 BGRA resize->960x540 INTER_AREA (exact 2x2 box)"),
    # BASELINE configs[2]: resize=960,540 (AREA, exact 2x2) -> filter-rotate=90 -> configured watermark 256x64 r,b,16,16 @60
    # a GIF-album style batch: filter-gotham (HSV modulate + colorize + gamma + contrast, 6 CPU sweeps) fused, in place
    "gotham": (1920, 1080, 1920, 1080, 256, -2, 2 * 1920 * 1080 * 4,
               "batch 256 of 1920x1080 BGRA filter-gotham in place (one fused pointwise launch)"),
    "gamma": (1920, 1080, 1920, 1080, 256, -3, 2 * 1920 * 1080 * 4,
              "batch 256 of 1920x1080 BGRA filter-gamma=2.2 in place (LUT)"),
    "chain": (1920, 1080, 540, 960, 1024, -1, 1920 * 1080 * 4 + 540 * 960 * 4,
              "batch 1024 of 1920x1080 BGRA resize(960x540)+rotate(90)+watermark alpha-blend chain"),
    # SURVEY 8(d) cfg3's second variant: the same chain with resize=224,224 (general AREA kernel, then rotate, then blend)
    "chain224": (1920, 1080, 224, 224, 1024, -1, 1920 * 1080 * 4 + 224 * 224 * 4,
                 "batch 1024 of 1920x1080 BGRA resize(224x224)+rotate(90)+watermark alpha-blend chain"),
}


def mixed_resident(imp, n, steps, warmup, rank, world, c=4):
    """BASELINE configs[4] without PCIe: this rank's share of n mixed-size BGRA frames (workloads.mixed_sizes) resident in
    HBM, every one resized to 224 wide (INTER_AREA, bridge.c:190).  A step = all of them once: (a) one
    impgpu_batch_resize_mixed call, (b) one impgpu_batch_cv_resize launch per frame from one thread."""
    import torch
    from ngx_http_imgproc_amd.workloads import MIXED_RESIZE, mixed_sizes

    sizes = mixed_sizes(n)[rank::world]
    cfg = imp.Config()
    g = torch.Generator(device="cuda")
    g.manual_seed(0x1A4D0005 + rank)
    srcs, dsts, items, alg = [], [], [], 0
    for w, h in sizes:
        rc, (dw, dh, _) = imp.resize_geometry(w, h, MIXED_RESIZE.decode(), cfg)
        assert rc == 0
        sp, dp = (w * c + 3) & ~3, (dw * c + 3) & ~3                 # cvCreateImage's widthStep
        srcs.append(torch.randint(0, 256, (h, sp), dtype=torch.uint8, device="cuda", generator=g))
        dsts.append(torch.zeros((dh, dp), dtype=torch.uint8, device="cuda"))
        items.append((srcs[-1].data_ptr(), w, h, sp, dsts[-1].data_ptr(), dw, dh, dp))
        alg += w * h * c + dw * dh * c
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    arr = (imp.ResizeItem * len(items))(*[imp.ResizeItem(*it) for it in items])

    def gathered():
        assert imp.lib.impgpu_batch_resize_mixed(arr, len(items), c, 0, stream.cuda_stream) == 0

    def per_frame():
        for sp, w, h, ss, dp, dw, dh, ds in items:
            imp.batch_cv_resize(sp, 0, w, h, ss, dp, 0, dw, dh, ds, c, 1, imp.INTER_AREA, stream=stream.cuda_stream)

    res = {}
    for name, fn in (("one_call", gathered), ("launch_per_frame", per_frame)):
        for _ in range(max(1, warmup // 10)):
            fn()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(steps):
            fn()
        ev1.record(stream)
        host = time.perf_counter() - t0            # the calls return after enqueue: what the host spent issuing them
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        secs = max(wall, ev0.elapsed_time(ev1) / 1e3)
        res[name] = {"host_ms_per_step": round(host / steps * 1e3, 3),"images_per_sec": round(len(items) * steps / secs, 1), "ms_per_step": round(secs / steps * 1e3, 3),
                     "alg_GBps": round(alg * steps / secs / 1e9, 1), "frac_of_8TBps": round(alg * steps / secs / 1e9 / HBM_PEAK_GBPS, 4),
                     "device_ms_per_step": round(ev0.elapsed_time(ev1) / steps, 3)}
    return {"metric": "images/sec mixed-size resident stream resize=224,0", "unit": "images/sec", "value": res["one_call"]["images_per_sec"],
            "rank": rank, "n_gpus": world, "frames_this_rank": len(items), "channels": c, "steps": steps, "dtype": "u8",
            "data": "synthetic (seeded sizes, torch.randint frames, device-resident)",
            "source_GB_this_rank": round(sum(w * h * c for w, h in sizes) / 1e9, 2), **res,
            "config": {"workload": "BASELINE configs[4] device-resident: %d frames, long side log-uniform 256..3840, resize=224,0 (INTER_AREA)" % n}}


def e2e(imp, n_requests, pinned):
    """PCIe-inclusive request loop on one stream: upload 1080p frame -> cubic resize -> download 224x224.
    Never the headline `value`; reported in DESIGN.md next to the device-resident number."""
    import ctypes as C
    import numpy as np

    rng = np.random.Generator(np.random.PCG64(0x1A4D0001))
    frame = rng.integers(0, 256, size=(1080, 1920, 4), dtype=np.uint8)
    out = np.empty((224, 224, 4), np.uint8)
    lib = imp.lib
    if pinned:
        hsrc = lib.impgpu_host_alloc(frame.nbytes)
        hdst = lib.impgpu_host_alloc(out.nbytes)
        C.memmove(hsrc, frame.ctypes.data, frame.nbytes)

    def one():
        h = C.c_void_p()
        if pinned:
            rc = lib.impgpu_image_upload_pinned(hsrc, 1920, 1080, 4, 1920 * 4, C.byref(h))
        else:
            rc = lib.impgpu_image_upload(frame.ctypes.data, 1920, 1080, 4, 1920 * 4, C.byref(h))
        assert rc == 0
        assert lib.impgpu_cv_resize(C.byref(h), 224, 224, imp.INTER_CUBIC) == 0
        if pinned:
            assert lib.impgpu_image_download_pinned(h, hdst, 224 * 4) == 0
            assert lib.impgpu_sync() == 0
        else:
            assert lib.impgpu_image_download(h, out.ctypes.data, 224 * 4) == 0
        lib.impgpu_image_release(C.byref(h))

    for _ in range(8):
        one()
    t0 = time.perf_counter()
    for _ in range(n_requests):
        one()
    dt = time.perf_counter() - t0
    if pinned:
        lib.impgpu_host_free(hsrc)
        lib.impgpu_host_free(hdst)
    return n_requests / dt


# the modes whose CPU leg is one cvResize: source (h, w), destination (w, h), interpolation
CPU_RESIZE_MODES = {"cubic": ((1080, 1920), (224, 224), "INTER_CUBIC"), "area": ((1080, 1920), (224, 224), "INTER_AREA"),
                    "lanczos": ((2160, 3840), (1920, 1080), "INTER_LANCZOS4")}


def cpu_baseline(seconds_budget=12.0, mode="cubic"):
    """The oracle on the host's cores, bounded sample: cv_resize of the mode's geometry (CUBIC on 1080p BGRA frames for the
    headline; AREA and the 4K LANCZOS4 of cfg4 for their modes; every other mode reports the headline's and says so).

    Timed on oracle/liboracle_fast.so -- the oracle's sources built `-O3 -march=native` on this host, float
    contraction still off -- after checking that it returns the same bytes as the `-O2` checker build the tests use."""
    import ctypes as C
    import subprocess
    import numpy as np
    import oracle_lib as orc

    flags = "-O2 (checker build)"
    fast = None
    try:
        subprocess.check_call(["make", "-s", "-C", orc.ORACLE_DIR, "liboracle_fast.so"])
        fast = C.CDLL(os.path.join(orc.ORACLE_DIR, "liboracle_fast.so"))
        fast.orc_image_from.restype = C.c_void_p
        fast.orc_image_from.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        fast.orc_image_create.restype = C.c_void_p
        fast.orc_image_create.argtypes = [C.c_int] * 3
        fast.orc_image_data.restype = C.c_void_p
        fast.orc_image_data.argtypes = [C.c_void_p]
        fast.orc_cv_resize.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        flags = "-O3 -march=native -ffp-contract=off"
    except Exception:       # no compiler on this host: time the checker build and say so
        fast = None

    matches = mode in CPU_RESIZE_MODES
    (sh, sw), (dw, dh), interp_name = CPU_RESIZE_MODES[mode if matches else "cubic"]
    interp = getattr(orc, interp_name)
    rng = np.random.Generator(np.random.PCG64(0x1A4D0001))
    arrays = [rng.integers(0, 256, size=(sh, sw, 4), dtype=np.uint8) for _ in range(4)]
    if fast is not None:
        lib = fast
        frames = [C.c_void_p(lib.orc_image_from(a.ctypes.data, sw, sh, 4, sw * 4)) for a in arrays]
        new_dst = lambda: C.c_void_p(lib.orc_image_create(dw, dh, 4))
        dst = new_dst()
        for a, f in zip(arrays[:1 if sh > 1080 else 4], frames):       # same bytes as the checker build, or the number means nothing
            lib.orc_cv_resize(f, dst, interp)
            got = np.ctypeslib.as_array((C.c_uint8 * (dw * dh * 4)).from_address(lib.orc_image_data(dst))).reshape(dh, dw, 4)
            assert np.array_equal(got, orc.cv_resize(a, dw, dh, interp)), "liboracle_fast.so differs from liboracle.so"
    else:
        lib = orc.lib
        imgs = [orc.Img(a) for a in arrays]
        frames = [im.h for im in imgs]
        new_dst = lambda: C.c_void_p(lib.orc_image_create(dw, dh, 4))
        dst = new_dst()
    n = 0
    t0 = time.perf_counter()
    while True:
        for f in frames:
            lib.orc_cv_resize(f, dst, interp)
        n += len(frames)
        dt = time.perf_counter() - t0
        if dt >= seconds_budget or n >= 4096:
            break
    # SURVEY 8(d) also asks for the all-cores figure (= nginx worker_processes N): independent workers over the same
    # frames, one thread per host core (ctypes drops the GIL inside the call), a further ~6 s
    import threading

    open_cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    cores = min(open_cores, int(os.environ.get("IMPGPU_BENCH_CPU_THREADS", "16")))   # a one-GPU box's CPU share is 16 cores
    counts = [0] * cores
    stop_at = time.perf_counter() + 6.0

    def worker(i):
        out = new_dst()
        k = 0
        while time.perf_counter() < stop_at:
            lib.orc_cv_resize(frames[k % len(frames)], out, interp)
            k += 1
        counts[i] = k

    t1 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(cores)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt_all = time.perf_counter() - t1
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": round(n / dt, 2),
        "unit": "images/sec",
        "cores": 1,
        "kind": "port",
        "build": "gcc " + flags + " (oracle/Makefile); bytes checked equal to the -O2 checker build on this sample",
        "cpu_model": cpu_model,
        "sample": "%d frames %dx%d BGRA -> %dx%d %s via the oracle (OpenCV 2.4.9 semantics restated in C), "
                  "single thread, %.1f s" % (n, sw, sh, dw, dh, interp_name, dt),
        "workload_matches_mode": matches,
        # NOT every core of the host: `threads` independent workers (one GPU's share of the box), out of `host_cores` this
        # process may run on and `host_cpus` the machine has
        "multi_thread": {"value": round(sum(counts) / dt_all, 2), "threads": cores, "host_cores": open_cores, "host_cpus": os.cpu_count(),
                         "sample": "%d frames, %d independent worker threads, %.1f s" % (sum(counts), cores, dt_all)},
    }


def request_stream(imp, n_requests, n_threads, queue_depth, inflight, rank=0, world=1, c=4):
    """BASELINE configs[4]: mixed-size request stream (ngx_http_imgproc_amd.workloads.mixed_sizes), each request =
    upload (pinned) -> resize=224,0 (keep aspect; AREA, what the reference runs, bridge.c:588-604 on whatever size
    arrives) -> download.  Requests shard round-robin over ranks (request i -> rank i mod world, no collective); on a
    rank they wait in a queue of `queue_depth` entries from which `n_threads` host threads -- each with its own lane
    = HIP stream -- pull; a thread keeps up to `inflight` requests enqueued on its stream before it waits for them."""
    import ctypes as C
    import queue
    import threading
    import numpy as np
    from ngx_http_imgproc_amd.shard import round_robin
    from ngx_http_imgproc_amd.workloads import MIXED_RESIZE, mixed_sizes

    all_sizes = mixed_sizes(n_requests)
    mine = [all_sizes[i] for i in round_robin(n_requests, rank, world)]
    lib = imp.lib
    bound = NUMA_BIND and lib.impgpu_env_bind_thread() == 0        # (the pinned source below is then first touched on the device's node)
    # one pinned source buffer (largest frame) filled with noise: every request reads its w*h*4 prefix
    maxpx = max(w * h for w, h in all_sizes)
    hsrc = lib.impgpu_host_alloc(maxpx * c + 64)
    rng = np.random.Generator(np.random.PCG64(0x1A4D0005))
    noise = rng.integers(0, 256, size=maxpx * c, dtype=np.uint8)
    C.memmove(hsrc, noise.ctypes.data, noise.nbytes)
    cfg = imp.Config()
    pending = queue.Queue(maxsize=max(1, queue_depth))
    errors = []
    out_bytes = 224 * 224 * 4 * 4          # the short side is at most 224 * 16/9 (sized for 4 channels)

    def feeder():
        for item in mine:
            pending.put(item)
        for _ in range(n_threads):
            pending.put(None)

    def worker():
        if NUMA_BIND:
            lib.impgpu_env_bind_thread()                          # host threads on the GPU's NUMA node
        hdst = lib.impgpu_host_alloc(out_bytes * inflight)
        live = []
        done = False
        while not done:
            item = pending.get()
            if item is None:
                done = True
            else:
                w, h = item
                img = C.c_void_p()
                rc = lib.impgpu_image_upload_pinned(hsrc, w, h, c, w * c, C.byref(img))
                if rc == 0:
                    rc = lib.impgpu_resize(C.byref(img), MIXED_RESIZE, C.byref(cfg.c), 0)
                if rc == 0:
                    ow = lib.impgpu_image_width(img)
                    rc = lib.impgpu_image_download_pinned(img, hdst + out_bytes * len(live), (ow * c + 3) & ~3)
                live.append(img)
                if rc:
                    errors.append((item, rc))
                    done = True
            if live and (done or len(live) >= inflight):
                if lib.impgpu_sync():
                    errors.append(("sync", 90))
                    done = True
                for img in live:
                    lib.impgpu_image_release(C.byref(img))
                live = []
        lib.impgpu_host_free(hdst)

    threads = [threading.Thread(target=worker) for _ in range(n_threads)]
    feed = threading.Thread(target=feeder)
    t0 = time.perf_counter()
    feed.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    if errors:      # unblock the feeder before failing
        try:
            while True:
                pending.get_nowait()
        except queue.Empty:
            pass
    feed.join(timeout=5)
    lib.impgpu_host_free(hsrc)
    if errors:
        raise SystemExit("request_stream failed: %r" % errors[:3])
    return {"requests": len(mine), "seconds": dt, "source_bytes": sum(w * h * c for w, h in mine),
            "numa_node": lib.impgpu_env_numa_node(), "threads_bound": bool(bound)}


NUMA_BIND = True        # --no-numa clears it: --stream worker threads (and their pinned buffers) on the device's NUMA node


def jpeg_pool(n_files):
    """The first n_files sizes of the mixed-size stream as photograph-like quality-90 4:2:0 JPEGs (Pillow encodes them;
    ngx_http_imgproc_amd.workloads.photo_like is the content, cut out of one 4K frame).  Requests cycle through this pool:
    the library caches nothing per file, so a repeated file costs what a new one costs."""
    import io
    from PIL import Image
    from ngx_http_imgproc_amd.workloads import mixed_sizes, photo_like

    sizes = mixed_sizes(n_files)
    side = max(max(w, h) for w, h in sizes)
    big = photo_like(side, side, seed=5)
    files = []
    for k, (w, h) in enumerate(sizes):
        y0, x0 = (k * 37) % (side - h + 1), (k * 91) % (side - w + 1)
        b = io.BytesIO()
        Image.fromarray(big[y0:y0 + h, x0:x0 + w]).save(b, "JPEG", quality=90, subsampling="4:2:0")
        files.append((w, h, b.getvalue()))
    return files


def jpeg_stream(imp, n_requests, n_threads, queue_depth, decoder, files, rank=0, world=1, batch=1, out_quality=0):
    """The same request stream with the requests arriving as what they are in production -- JPEG files (bridge.c:376-378,
    :545-552): decode -> resize=224,0 -> download.  decoder = "device": impgpu_image_decode_jpeg (the compressed bytes cross
    the link; Huffman, IDCT, upsampling, colour on the device); "hosthuff": the same with the entropy stage on the calling
    thread (IMPGPU_JPEG_HUFF=host); "host": the reference's structure -- libjpeg-turbo on the calling thread (Pillow's,
    which releases the GIL) and impgpu_image_upload of the decoded frame.  out_quality > 0: the answer is a JPEG file too
    (cvEncodeImage at bridge.c:704 with that quality): written on the device (impgpu_batch_encode_jpeg) and only the file is
    downloaded -- or, for decoder "host", by libjpeg-turbo on the calling thread from the downloaded thumbnail."""
    import ctypes as C
    import io
    import queue
    import threading
    import numpy as np
    from ngx_http_imgproc_amd import ResizeItem
    from ngx_http_imgproc_amd.shard import round_robin
    from ngx_http_imgproc_amd.workloads import MIXED_RESIZE

    lib = imp.lib
    mine = [files[i % len(files)] for i in round_robin(n_requests, rank, world)]
    os.environ["IMPGPU_JPEG_HUFF"] = "host" if decoder == "hosthuff" else "device"
    cfg = imp.Config()
    pending = queue.Queue(maxsize=max(1, queue_depth))
    errors = []
    out_bytes = 224 * 224 * 4 * 4
    answer_bytes = [0]
    enc_cap = lib.impgpu_jpeg_encode_bound(224, 224, 3)

    def feeder():
        for item in mine:
            pending.put(item)
        for _ in range(n_threads):
            pending.put(None)

    bound_ok = []

    def worker():
        if NUMA_BIND:
            bound_ok.append(lib.impgpu_env_bind_thread() == 0)
        hdst = lib.impgpu_host_alloc(out_bytes * max(1, batch))
        if decoder == "host":
            from PIL import Image
        nb = max(1, batch)
        files_out = np.empty((nb, enc_cap), dtype=np.uint8) if out_quality and decoder != "host" else None
        if files_out is not None:
            e_outs = (C.c_void_p * nb)(*[files_out[k].ctypes.data for k in range(nb)])
            e_caps = (C.c_size_t * nb)(*[enc_cap] * nb)
            e_lens = (C.c_size_t * nb)()
            e_codes = (C.c_int * nb)()
        sent = 0

        def host_encode(k, ow, oh):                                  # the reference's encoder on the downloaded thumbnail
            step = (ow * 3 + 3) & ~3
            a = np.ctypeslib.as_array(C.cast(hdst + out_bytes * k, C.POINTER(C.c_ubyte)), shape=(oh * step,)).reshape(oh, step)[:, : ow * 3].reshape(oh, ow, 3)
            b = io.BytesIO()
            Image.fromarray(a).save(b, format="JPEG", quality=out_quality, subsampling=2)
            return len(b.getvalue())

        done = False
        while not done:
            items = []
            while len(items) < max(1, batch):
                try:
                    item = pending.get(block=not items)          # wait for the first, take what else is already there
                except queue.Empty:
                    break
                if item is None:
                    done = True
                    break
                items.append(item)
            if not items:
                break
            n = len(items)
            imgs = (C.c_void_p * n)()
            rc = 0
            if decoder == "host":
                for k, (w, h, blob) in enumerate(items):
                    a = np.asarray(Image.open(io.BytesIO(blob)))
                    one = C.c_void_p()
                    rc = rc or lib.impgpu_image_upload(a.ctypes.data, w, h, 3, w * 3, C.byref(one))
                    imgs[k] = one
            else:
                blobs = (C.c_char_p * n)(*[b for _, _, b in items])
                sizes = (C.c_size_t * n)(*[len(b) for _, _, b in items])
                codes = (C.c_int * n)()
                rc = lib.impgpu_batch_decode_jpeg(blobs, sizes, n, imgs, codes)
                rc = rc or max(codes)
            if n > 1 and rc == 0:
                # the decoded frames of the batch are resized together too: impgpu_batch_resize_mixed, one descriptor launch
                # (the per-frame Resize() loop of bridge.c:588-604 over whatever sizes arrived), then one download each
                outs = (C.c_void_p * n)()
                its = (ResizeItem * n)()
                ow_, oh_, ip_ = C.c_int(), C.c_int(), C.c_int()
                for k in range(n):
                    one = C.c_void_p(imgs[k])
                    sw_, sh_ = lib.impgpu_image_width(one), lib.impgpu_image_height(one)
                    rc = rc or lib.impgpu_resize_geometry(sw_, sh_, MIXED_RESIZE, C.byref(cfg.c), 0, ow_, oh_, ip_)
                    o = C.c_void_p()
                    rc = rc or lib.impgpu_image_create(ow_.value, oh_.value, 3, C.byref(o))
                    outs[k] = o
                    if rc == 0:
                        its[k] = ResizeItem(lib.impgpu_image_device_ptr(one), sw_, sh_, lib.impgpu_image_step(one),
                                            lib.impgpu_image_device_ptr(o), ow_.value, oh_.value, lib.impgpu_image_step(o))
                if rc == 0:
                    rc = lib.impgpu_batch_resize_mixed(its, n, 3, 0, None)
                if files_out is not None:
                    if rc == 0:
                        rc = lib.impgpu_batch_encode_jpeg(outs, n, out_quality, e_outs, e_caps, e_lens, e_codes)
                        rc = rc or max(e_codes[:n])
                        sent += sum(e_lens[:n])
                else:
                    for k in range(n):
                        o = C.c_void_p(outs[k])
                        if rc == 0:
                            rc = lib.impgpu_image_download_pinned(o, hdst + out_bytes * k, lib.impgpu_image_step(o))
                    if rc == 0:
                        rc = lib.impgpu_sync()
                    for k in range(n):
                        o = C.c_void_p(outs[k])
                        if rc == 0 and out_quality:
                            sent += host_encode(k, lib.impgpu_image_width(o), lib.impgpu_image_height(o))
                        elif rc == 0:
                            sent += lib.impgpu_image_step(o) * lib.impgpu_image_height(o)
                for k in range(n):
                    o = C.c_void_p(outs[k])
                    if o:
                        lib.impgpu_image_release(C.byref(o))
            else:
                for k in range(n):
                    one = C.c_void_p(imgs[k])
                    if rc == 0:
                        rc = lib.impgpu_resize(C.byref(one), MIXED_RESIZE, C.byref(cfg.c), 0)
                    if rc == 0 and files_out is not None:
                        rc = lib.impgpu_image_encode_jpeg(one, out_quality, e_outs[0], enc_cap, e_lens)
                        sent += e_lens[0]
                    elif rc == 0:
                        ow = lib.impgpu_image_width(one)
                        rc = lib.impgpu_image_download_pinned(one, hdst + out_bytes * k, (ow * 3 + 3) & ~3)
                    imgs[k] = one
                if rc == 0 and files_out is None:
                    rc = lib.impgpu_sync()
                    for k in range(n):
                        one = C.c_void_p(imgs[k])
                        ow, oh = lib.impgpu_image_width(one), lib.impgpu_image_height(one)
                        sent += host_encode(k, ow, oh) if out_quality else ((ow * 3 + 3) & ~3) * oh
            for k in range(n):
                one = C.c_void_p(imgs[k])
                if one:
                    lib.impgpu_image_release(C.byref(one))
            if rc:
                errors.append(((items[0][0], items[0][1]), rc))
                break
        lib.impgpu_host_free(hdst)
        answer_bytes[0] += sent                                      # (under the GIL)

    threads = [threading.Thread(target=worker) for _ in range(n_threads)]
    feed = threading.Thread(target=feeder, daemon=True)
    t0 = time.perf_counter()
    feed.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    if errors:
        raise SystemExit("jpeg_stream failed: %r" % errors[:3])
    return {"requests": len(mine), "seconds": dt, "source_bytes": sum(w * h * 3 for w, h, _ in mine),
            "file_bytes": sum(len(b) for _, _, b in mine), "answer_bytes": answer_bytes[0], "numa_node": lib.impgpu_env_numa_node(),
            "threads_bound": bool(NUMA_BIND and bound_ok and all(bound_ok))}


def jpeg_stream_native(n_requests, n_threads, files, batch, out_quality, device):
    """The same stream driven by C threads over the C ABI (tests/c/stream_harness.c) -- what a pool of nginx workers is.  The
    Python driver above holds the interpreter's lock for a millisecond per batch and hands it over in 5 ms slices: with more
    than two threads it measures that (device idle 45-50 % of the time, tools/trace_busy.py), not the library."""
    import struct
    import subprocess
    import tempfile

    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "c", "_build", "stream_harness")
    if not os.path.exists(exe):
        raise SystemExit("tests/c/_build/stream_harness is not built (python -c 'import __graft_entry__ as g; g.build()')")
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(struct.pack("<I", len(files)))
        for _, _, b in files:
            f.write(struct.pack("<I", len(b)))
            f.write(b)
        pool = f.name
    try:
        warm = max(2048, 8 * n_threads * batch)
        env = dict(os.environ, IMPGPU_DEVICE=str(device))
        out = subprocess.run([exe, pool, str(n_requests), str(n_threads), str(batch), str(out_quality), str(warm)], env=env, capture_output=True, text=True)
        if out.returncode:
            raise SystemExit("stream_harness failed (%d): %s" % (out.returncode, out.stderr[-400:]))
        r = json.loads(out.stdout.strip().splitlines()[-1])
    finally:
        os.unlink(pool)
    src = 0
    for i in range(n_requests):
        w, h, _ = files[i % len(files)]
        src += w * h * 3
    return {"requests": r["requests"], "seconds": r["seconds"], "source_bytes": src, "file_bytes": r["file_bytes"], "answer_bytes": r["answer_bytes"],
            "numa_node": r["numa_node"], "threads_bound": bool(r.get("threads_bound"))}


def jpeg_stage_profile(imp, files, batch, out_quality, reps=6):
    """Where a batch's time goes, one thread, nothing else on the device: the decode call's own stages
    (impgpu_jpeg_stage_times: host clock and stream events), then the resize, the encoder / the download timed around their
    calls.  Microseconds per call of `batch` files; the median of `reps`."""
    import ctypes as C
    import numpy as np
    from ngx_http_imgproc_amd import ResizeItem
    from ngx_http_imgproc_amd.workloads import MIXED_RESIZE

    lib = imp.lib
    n = min(batch, len(files)) if batch > 1 else 1
    cfg = imp.Config()
    whole_before = os.environ.get("IMPGPU_JPEG_WHOLE")
    os.environ["IMPGPU_JPEG_WHOLE"] = "1"          # the stages of ONE launch per kernel (the decode call otherwise cuts a batch in two)
    items = [files[i % len(files)] for i in range(n)]
    blobs = (C.c_char_p * n)(*[b for _, _, b in items])
    sizes = (C.c_size_t * n)(*[len(b) for _, _, b in items])
    enc_cap = lib.impgpu_jpeg_encode_bound(224, 224, 3)
    host = np.empty((n, max(enc_cap, 224 * 224 * 16)), dtype=np.uint8)
    datas = (C.c_void_p * n)(*[host[k].ctypes.data for k in range(n)])
    caps = (C.c_size_t * n)(*[enc_cap] * n)
    lens = (C.c_size_t * n)()
    codes = (C.c_int * n)()
    steps = (C.c_int * n)()
    names = ["headers", "unstuff", "job_tables", "enqueue", "wait", "upload", "walks", "mend", "select", "write", "dcfix", "pixels"]
    rows = []
    lib.impgpu_jpeg_profile(1)
    try:
        for _ in range(reps + 2):
            imgs = (C.c_void_p * n)()
            outs = (C.c_void_p * n)()
            assert lib.impgpu_batch_decode_jpeg(blobs, sizes, n, imgs, codes) == 0 and not any(codes)
            st = (C.c_double * 16)()
            lib.impgpu_jpeg_stage_times(st, 16)
            its = (ResizeItem * n)()
            ow, oh, ip = C.c_int(), C.c_int(), C.c_int()
            for k in range(n):
                one = C.c_void_p(imgs[k])
                sw_, sh_ = lib.impgpu_image_width(one), lib.impgpu_image_height(one)
                lib.impgpu_resize_geometry(sw_, sh_, MIXED_RESIZE, C.byref(cfg.c), 0, ow, oh, ip)
                o = C.c_void_p()
                assert lib.impgpu_image_create(ow.value, oh.value, 3, C.byref(o)) == 0
                outs[k] = o
                steps[k] = lib.impgpu_image_step(o)
                its[k] = ResizeItem(lib.impgpu_image_device_ptr(one), sw_, sh_, lib.impgpu_image_step(one), lib.impgpu_image_device_ptr(o), ow.value, oh.value, steps[k])
            lib.impgpu_sync()
            t0 = time.perf_counter()
            assert lib.impgpu_batch_resize_mixed(its, n, 3, 0, None) == 0
            lib.impgpu_sync()
            t1 = time.perf_counter()
            if out_quality:
                assert lib.impgpu_batch_encode_jpeg(outs, n, out_quality, datas, caps, lens, codes) == 0
            else:
                assert lib.impgpu_batch_download(outs, n, datas, steps) == 0
            t2 = time.perf_counter()
            for k in range(n):
                for arr in (imgs, outs):
                    one = C.c_void_p(arr[k])
                    lib.impgpu_image_release(C.byref(one))
            rows.append(list(st)[:12] + [(t1 - t0) * 1e6, (t2 - t1) * 1e6])
    finally:
        lib.impgpu_jpeg_profile(0)
    med = np.median(np.array(rows[2:]), axis=0)
    out = {nm: round(float(v), 1) for nm, v in zip(names, med[:12])}
    out["entropy"] = round(float(sum(med[6:11])), 1)
    out["resize"] = round(float(med[12]), 1)
    out["encode" if out_quality else "download"] = round(float(med[13]), 1)
    out["files_per_call"] = n
    if whole_before is None:
        os.environ.pop("IMPGPU_JPEG_WHOLE", None)
    else:
        os.environ["IMPGPU_JPEG_WHOLE"] = whole_before
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--prewarm-sec", type=float, default=0.5, help="untimed device pre-warm before the warm-up steps")
    ap.add_argument("--warmup", type=int, default=20)      # ~25 ms: past the clock ramp of a cold device
    ap.add_argument("--mode", default="cubic", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override frames per step (default: workload's)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--stream", type=int, default=0, metavar="N",
                    help="instead of the headline, run N mixed-size PCIe-inclusive requests (cfg5) over --threads host threads")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--queue-depth", type=int, default=4096, help="--stream: requests waiting in the rank's queue")
    ap.add_argument("--inflight", type=int, default=4, help="--stream: requests a thread enqueues before it waits")
    ap.add_argument("--no-numa", action="store_true", help="--stream: do not bind the worker threads to the device's NUMA node")
    ap.add_argument("--jpeg", default="", choices=("", "device", "hosthuff", "host", "all"),
                    help="--stream: the requests arrive as JPEG files; where they are decoded (all = the three one after the other)")
    ap.add_argument("--jpeg-batch", type=int, default=1, help="--stream --jpeg: requests a thread takes from the queue and decodes with one impgpu_batch_decode_jpeg call")
    ap.add_argument("--jpeg-out", type=int, default=0, metavar="Q",
                    help="--stream --jpeg: the answers are JPEG files of quality Q too (the module's default is 86), encoded where the request was decoded")
    ap.add_argument("--jpeg-files", type=int, default=64, help="--stream --jpeg: distinct files the requests cycle through")
    ap.add_argument("--native", action="store_true", help="--stream --jpeg device: the request threads are C threads (tests/c/stream_harness.c), not Python's")
    ap.add_argument("--workers", type=int, default=0, metavar="N",
                    help="the reference's process model: N worker PROCESSES (tests/c/worker_harness.c), one synchronous request each "
                         "(JPEG in -> resize=224,0 -> JPEG out), through impgpu_broker (--broker-lanes) or, with --in-process, each with "
                         "libimpgpu.so linked in (at most 6 on a pool box); --seconds per point")
    ap.add_argument("--in-process", action="store_true")
    ap.add_argument("--broker-lanes", type=int, default=4)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--mixed", type=int, default=0, metavar="N",
                    help="BASELINE configs[4] with the N frames already in HBM: resize=224,0 over mixed sizes, one "
                         "impgpu_batch_resize_mixed call per step (and, for comparison, one launch per frame)")
    ap.add_argument("--channels", type=int, default=4, choices=(3, 4), help="--mixed / --stream: BGRA (4) or BGR (3, what a JPEG decodes to)")
    ap.add_argument("--e2e", type=int, default=0, metavar="N",
                    help="instead of the headline, time N PCIe-inclusive requests (upload + resize + download) and exit")
    args = ap.parse_args()
    global NUMA_BIND
    NUMA_BIND = not args.no_numa

    if args.workers:
        # (before anything of this process touches the GPU: the workers and the broker are children, started here)
        import subprocess

        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import worker_scaling as ws

        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c")])
        pool = os.path.join(ROOT, "gpurun_out", "jpeg_pool.bin")
        os.makedirs(os.path.dirname(pool), exist_ok=True)
        if not os.path.exists(pool):                          # (made by a child: jpeg_pool imports nothing of the device, but torch is heavy)
            subprocess.check_call([sys.executable, "-c", "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import worker_scaling as w; w.make_pool(%r)"
                                   % (ROOT, os.path.join(ROOT, "tools"), pool)])
        name = "/impgpu-bench-%d" % os.getpid()
        broker = None if args.in_process else ws.start_broker(name, args.broker_lanes, 0)
        try:
            r = ws.run_point(pool, "direct" if args.in_process else "broker", args.workers, args.seconds, None, name)
        finally:
            if broker:
                ws.stop_broker(broker)
        print(json.dumps({
            "metric": "requests/sec, N worker processes, one synchronous request each: JPEG in -> resize=224,0 -> JPEG out", "value": r["requests_per_s"],
            "unit": "requests/sec", "n_gpus": 1, "higher_is_better": True, "dtype": "u8", "data": "synthetic (photograph-like quality-90 4:2:0 JPEG files, 256 px - 4K, Pillow-encoded)",
            "config": {"workload": "BASELINE configs[4] from the reference's process model (docs/02 - Configuration.md:18 worker_processes; module.c:298 RunJob synchronous)",
                       "workers": args.workers, "path": "libimpgpu.so in every worker" if args.in_process else "impgpu_broker, %d lanes" % args.broker_lanes,
                       "seconds": args.seconds},
            "latency_us": {"p50": r["p50_us"], "p95": r["p95_us"], "p99": r["p99_us"]}, "files_per_launch": r["mean_batch"],
            "chain_timeouts": r["chain_timeouts"], "vs_baseline": None}))
        return

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # IMPGPU_BENCH_FORCE_DIST=1 takes the N>1 code path (RCCL init, barriers, MAX over ranks) with a single rank:
    # the rehearsal of that path that a one-GPU box allows
    use_dist = world > 1 or (os.environ.get("IMPGPU_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"      # the pool exports VERSION: RCCL's banner would share stdout with the one JSON line
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    lib_path = os.path.join(ROOT, "ngx_http_imgproc_amd", "libimpgpu.so")
    if not os.path.exists(lib_path) and local_rank == 0:     # normally built by __graft_entry__.build(); hipcc is on the box
        import subprocess

        subprocess.check_call([sys.executable, os.path.join(ROOT, "ngx_http_imgproc_amd", "build.py")])
        subprocess.call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    if use_dist:
        dist.barrier(device_ids=[local_rank])
    import ngx_http_imgproc_amd as imp
    from ngx_http_imgproc_amd.shard import elapsed_max, job_totals

    imp.env_start(local_rank)
    if args.stream and args.jpeg:
        files = jpeg_pool(args.jpeg_files)
        lines = []
        for decoder in (("device", "hosthuff", "host") if args.jpeg == "all" else (args.jpeg,)):
            # untimed prefix: every lane (= thread) meets the common buffer sizes once, so that the timed part measures the
            # steady state of a server, not hipMalloc / hipHostMalloc
            native = args.native and decoder == "device"
            stages = jpeg_stage_profile(imp, files, args.jpeg_batch, args.jpeg_out) if decoder == "device" else None
            if not native:
                jpeg_stream(imp, min(args.stream, max(256, 24 * args.threads * args.jpeg_batch)), args.threads, args.queue_depth, decoder, files, rank, world, args.jpeg_batch, args.jpeg_out)
            if use_dist:
                dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize()
            if native:
                share = len(range(rank, args.stream, world))
                r = jpeg_stream_native(share, args.threads, files, args.jpeg_batch, args.jpeg_out, local_rank)
            else:
                r = jpeg_stream(imp, args.stream, args.threads, args.queue_depth, decoder, files, rank, world, args.jpeg_batch, args.jpeg_out)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier(device_ids=[local_rank])
            secs, (nreq, nbytes, fbytes, abytes) = job_totals(r["seconds"], [r["requests"], r["source_bytes"], r["file_bytes"], r["answer_bytes"]], dist if use_dist else None, "cuda")
            lines.append({
                "metric": "requests/sec, mixed-size JPEG request stream (256px-4K) decode + resize=224,0, PCIe-inclusive",
                "value": round(nreq / secs, 1), "unit": "requests/sec", "n_gpus": world, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "u8", "decoder": decoder,
                "data": "synthetic (seeded sizes, photograph-like content, quality-90 4:2:0 JPEG files, %d distinct)" % len(files),
                "compressed_MB_per_sec": round(fbytes / secs / 1e6, 1), "decoded_MB_per_sec": round(nbytes / secs / 1e6, 1),
                "bits_per_pixel": round(fbytes * 8 / (nbytes / 3), 2), "seconds": round(secs, 3),
                "answers": ("JPEG quality %d, encoded on the %s" % (args.jpeg_out, "host" if decoder == "host" else "device")) if args.jpeg_out else "raw B,G,R thumbnails",
                "answer_bytes_per_request": round(abytes / max(1.0, nreq), 1),
                "driver": "C threads over the C ABI (tests/c/stream_harness.c)" if native else "Python threads (ctypes)",
                # what the stream moves at the least: the file in, its coefficient planes once, the decoded pixels once, the answer out
                "roofline": (lambda alg: {"bound": "hbm", "achieved": round(alg / secs / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg / secs / 8e12, 4),
                                          "traffic": None, "algorithmic_bytes_per_request": round(alg / max(1.0, nreq)),
                                          "link": {"compressed_GB_per_s": round(fbytes / secs / 1e9, 2), "measured_link_GB_per_s": 55.0, "frac": round(fbytes / secs / 55e9, 3)}})(
                    fbytes + abytes + nbytes + sum(((w + 15) // 16) * ((h + 15) // 16) * 6 * 128 for w, h, _ in files) * (nreq / len(files))),
                "stages_us_per_call": stages,
                "config": {"workload": "BASELINE configs[4] as JPEG files: %d requests, long side log-uniform 256..3840, resize=224,0 (INTER_AREA)" % int(nreq),
                           "threads_per_gpu": args.threads, "files_per_decode_call": args.jpeg_batch, "host_cores": os.cpu_count(), "queue_depth": args.queue_depth,
                           "numa_node": r["numa_node"], "threads_bound_to_node": r["threads_bound"],
                           "sharding": "request i -> rank i mod N, no collective"}})
        if rank == 0:
            for ln in lines:
                print(json.dumps(ln), flush=True)
        imp.env_destroy()
        if use_dist:
            dist.destroy_process_group()
        return
    if args.stream:
        # warm this rank's lanes / pools / clocks on a short untimed prefix, then the timed stream
        request_stream(imp, min(args.stream, 64 * args.threads), args.threads, args.queue_depth, args.inflight, rank, world, args.channels)
        if use_dist:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()
        r = request_stream(imp, args.stream, args.threads, args.queue_depth, args.inflight, rank, world, args.channels)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[local_rank])
        secs, (nreq, nbytes) = job_totals(r["seconds"], [r["requests"], r["source_bytes"]], dist if use_dist else None, "cuda")
        if rank == 0:
            print(json.dumps({
                "metric": "requests/sec, mixed-size request stream (256px-4K) resize=224,0, PCIe-inclusive",
                "value": round(nreq / secs, 1), "unit": "requests/sec", "n_gpus": world, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "u8",
                "data": "synthetic (seeded sizes, noise %s frames in pinned host memory, rows tightly packed)" % ("BGRA" if args.channels == 4 else "BGR"),
                "source_MB_per_sec": round(nbytes / secs / 1e6, 1), "seconds": round(secs, 3),
                "config": {"workload": "BASELINE configs[4]: %d requests, long side log-uniform 256..3840, resize=224,0 (INTER_AREA)" % int(nreq),
                           "threads_per_gpu": args.threads, "queue_depth": args.queue_depth, "inflight_per_thread": args.inflight, "channels": args.channels,
                           "numa_node": r["numa_node"], "threads_bound_to_node": r["threads_bound"],
                           "sharding": "request i -> rank i mod N, no collective"}}), flush=True)
        imp.env_destroy()
        if use_dist:
            dist.destroy_process_group()
        return
    if args.mixed:
        if use_dist:
            dist.barrier(device_ids=[local_rank])
        res = mixed_resident(imp, args.mixed, args.steps, args.warmup, rank, world, args.channels)
        if use_dist:                               # frames shard round-robin: images add up, the slowest rank sets the time
            oc = res["one_call"]
            ms, (frames,) = job_totals(oc["ms_per_step"], [res["frames_this_rank"]], dist, "cuda")
            res["value"] = round(frames / (ms * 1e-3), 1)
            res["frames_all_ranks"] = int(frames)
            res["scaling"] = "strong"
        if rank == 0:
            print(json.dumps(res), flush=True)
        imp.env_destroy()
        if use_dist:
            dist.destroy_process_group()
        return
    if args.e2e:
        res = {"metric": "PCIe-inclusive requests/sec: upload 1920x1080 BGRA + INTER_CUBIC ->224x224 + download, one stream",
               "pageable_frame_via_staging": round(e2e(imp, args.e2e, False), 1),
               "pinned_frame_direct": round(e2e(imp, args.e2e, True), 1), "requests": args.e2e, "unit": "images/sec"}
        print(json.dumps(res), flush=True)
        imp.env_destroy()
        return
    sw, sh, dw, dh, batch, interp, alg_bytes, label = WORKLOADS[args.mode]
    if args.batch:
        batch = args.batch

    # synthetic frames, resident in HBM before timing (requests shard round-robin: this rank's share)
    g = torch.Generator(device="cuda")
    g.manual_seed(0x1A4D0001 + rank)
    src = torch.randint(0, 256, (batch, sh, sw, 4), dtype=torch.uint8, device="cuda", generator=g)
    dst = torch.zeros((batch, dh, dw, 4), dtype=torch.uint8, device="cuda")
    # a real (non-null) stream: the library launches on exactly this one, so the events below see the kernels
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())

    cfg = None
    if args.mode in ("chain", "chain224"):
        ov = torch.randint(0, 256, (64, 256, 4), dtype=torch.uint8, generator=torch.Generator().manual_seed(0x1A4D00FF))
        ov[:, :, 3] = torch.linspace(0, 255, 256).to(torch.uint8)[None, :]
        cfg = imp.Config()
        assert cfg.prepare_watermark(ov.numpy(), "r", "b", 16, 16, 60) == 0

    def step():
        if interp in (-2, -3):
            rc = imp.batch_filters(src.data_ptr(), sh * sw * 4, sw, sh, 4, sw * 4, batch,
                                   ["gotham=1"] if interp == -2 else ["gamma=2.2"], 1, stream=stream.cuda_stream)
            assert rc == 0, rc
            return
        if cfg is not None:
            imp.batch_resize_rotate_watermark(src.data_ptr(), sh * sw * 4, sw, sh, sw * 4, dst.data_ptr(), dh * dw * 4, dw * 4,
                                              dh, dw, 90, cfg, 4, batch, stream=stream.cuda_stream)   # resize to dh x dw, then the quarter turn
            return
        imp.batch_cv_resize(src.data_ptr(), sh * sw * 4, sw, sh, sw * 4, dst.data_ptr(), dh * dw * 4, dw, dh, dw * 4,
                            4, batch, interp, stream=stream.cuda_stream)

    # device pre-warm: the first process on a cold MI355X reads ~5 % low for its first few hundred ms (clock ramp), which
    # W = 20 steps (25 ms) do not cover; spin the same step for --prewarm-sec before the W counted warm-up steps
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_sec:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if use_dist:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    dev_ms = ev0.elapsed_time(ev1)          # HIP events on the launch stream, whole timed region
    elapsed = max(wall, dev_ms / 1e3)
    # whole job: the slowest rank's time, every rank's frames (ngx_http_imgproc_amd.shard.job_totals; the same function under
    # two gloo ranks in tests/test_multirank.py)
    elapsed, (frames_all,) = job_totals(elapsed, [batch * args.steps], dist if use_dist else None, "cuda")
    dev_ms = elapsed_max(dev_ms, dist if use_dist else None, "cuda")

    if rank == 0:
        launch_ms = dev_ms / args.steps                  # one kernel launch per step (chain: its launches together)
        achieved = alg_bytes * batch / (launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.mode)
        if os.path.exists(tpath):                        # PMC pass result (rocprofv3 --pmc), per launch, corrected
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": "images/sec 1920x1080->224 bicubic resize" if args.mode == "cubic" else "images/sec " + args.mode,
            "value": round(frames_all / elapsed, 1),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic (torch.randint uint8 BGRA frames, seeded, device-resident)",
            "config": {"workload": label, "frames_per_step_per_gpu": batch, "sharding": "independent frames per rank, no collective",
                       "prewarm_sec": args.prewarm_sec},
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                # NOT measured in this run: the PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/pmc_probe.py,
                # separate passes, gfx950 FETCH x2 correction) is its own profiled run whose result is committed
                "traffic_source": ("profiles/traffic_%s.json" % args.mode) if traffic else None,
                "algorithmic_bytes_per_launch": alg_bytes * batch,
                "kernel_ms_per_launch": round(launch_ms, 4),
            },
        }
        if traffic and batch == json.load(open(tpath)).get("batch"):
            # what rocprofv3's FETCH_SIZE + WRITE_SIZE saw per launch, at this run's launch time
            out["roofline"]["traffic_rate_GBps"] = round(traffic / (launch_ms * 1e-3) / 1e9, 1)
        if args.mode == "cubic":
            # at a 34-byte tap pitch every 128-byte line of the 896 needed rows is touched, so the
            # least HBM can move for this access pattern is 896 rows x 7680 B + the destination
            lg = (896 * 7680 + dw * dh * 4) * batch
            out["roofline"]["line_granular"] = {"bytes_per_launch": lg, "achieved": round(lg / (launch_ms * 1e-3) / 1e9, 1),
                                                "frac": round(lg / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(mode=args.mode)
        print(json.dumps(out), flush=True)
    del src, dst
    imp.env_destroy()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
