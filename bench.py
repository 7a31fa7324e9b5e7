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
}


def cpu_baseline(seconds_budget=12.0):
    """Oracle cv_resize CUBIC on 1080p BGRA frames, one core, bounded sample."""
    import numpy as np
    import oracle_lib as orc

    rng = np.random.Generator(np.random.PCG64(0x1A4D0001))
    frames = [orc.Img(rng.integers(0, 256, size=(1080, 1920, 4), dtype=np.uint8)) for _ in range(4)]
    dst = orc.Img(handle=orc.lib.orc_image_create(224, 224, 4))
    orc.lib.orc_cv_resize(frames[0].h, dst.h, orc.INTER_CUBIC)   # warm
    n = 0
    t0 = time.perf_counter()
    while True:
        for f in frames:
            orc.lib.orc_cv_resize(f.h, dst.h, orc.INTER_CUBIC)
        n += len(frames)
        dt = time.perf_counter() - t0
        if dt >= seconds_budget or n >= 4096:
            break
    return {
        "value": round(n / dt, 2),
        "unit": "images/sec",
        "cores": 1,
        "kind": "port",
        "sample": "%d frames 1920x1080 BGRA -> 224x224 INTER_CUBIC via oracle/liboracle.so (OpenCV 2.4.9 "
                  "semantics restated in C, gcc -O2), single thread, %.1f s" % (n, dt),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="cubic", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override frames per step (default: workload's)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import ngx_http_imgproc_amd as imp

    imp.env_start(local_rank)
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

    def step():
        imp.batch_cv_resize(src.data_ptr(), sh * sw * 4, sw, sh, sw * 4, dst.data_ptr(), dh * dw * 4, dw, dh, dw * 4,
                            4, batch, interp, stream=stream.cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dev_ms = ev0.elapsed_time(ev1)          # HIP events on the launch stream, whole timed region
    elapsed = max(wall, dev_ms / 1e3)
    if world > 1:
        t = torch.tensor([elapsed, dev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_ms = float(t[0]), float(t[1])

    if rank == 0:
        launch_ms = dev_ms / args.steps                  # one kernel launch per step
        achieved = alg_bytes * batch / (launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.mode)
        if os.path.exists(tpath):                        # PMC pass result (rocprofv3 --pmc), per launch, corrected
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": "images/sec 1920x1080->224 bicubic resize" if args.mode == "cubic" else "images/sec " + args.mode,
            "value": round(world * batch * args.steps / elapsed, 1),
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
            "config": {"workload": label, "frames_per_step_per_gpu": batch, "sharding": "independent frames per rank, no collective"},
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes * batch,
                "kernel_ms_per_launch": round(launch_ms, 4),
            },
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    del src, dst
    imp.env_destroy()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
