"""N IMP worker PROCESSES on one GPU, one request at a time each (docs/02 - Configuration.md:18 worker_processes; module.c:298
RunJob synchronous): JPEG in -> resize=224,0 -> JPEG out, the mixed-size pool of bench.py --stream --jpeg.
    direct : every worker links libimpgpu.so and owns a device context (tests/c/worker_harness.c direct)
    broker : one impgpu_broker owns the device, workers are plain C clients of its shared-memory segment
Every size is warm before the clock; each point runs SECONDS (default 3).  One JSON line per point.
    python tools/worker_scaling.py direct 1 2 4 6
    python tools/worker_scaling.py broker 1 2 4 8 16 32 [--threads 2] [--gather-us 0] [--seconds 3]
(the GPU box allows at most 6 processes on the card: direct stops at 6, the broker is ONE such process however many workers)"""
import argparse
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HARNESS = os.path.join(ROOT, "tests", "c", "_build", "worker_harness")
BROKER = os.path.join(ROOT, "ngx_http_imgproc_amd", "impgpu_broker")


def write_pool(path, blobs):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(blobs)))
        for b in blobs:
            f.write(struct.pack("<I", len(b)))
            f.write(b)


def make_pool(path, n_files=64):
    if os.path.exists(path):
        return
    import bench

    write_pool(path, [b for _, _, b in bench.jpeg_pool(n_files)])


def start_broker(name, threads, gather_us, slots=64, extra=(), env=None):
    d = tempfile.mkdtemp(prefix="impb_")
    ready = os.path.join(d, "ready")
    p = subprocess.Popen([BROKER, "--name", name, "--threads", str(threads), "--gather-us", str(gather_us), "--slots", str(slots),
                          "--ready-file", ready] + list(extra), stderr=subprocess.PIPE, text=True, env=dict(os.environ, **(env or {})))
    t_end = time.time() + 180
    while not os.path.exists(ready):
        if p.poll() is not None or time.time() > t_end:
            raise SystemExit("broker did not start: %s" % p.stderr.read()[-800:])
        time.sleep(0.01)
    shutil.rmtree(d, ignore_errors=True)
    return p


def stop_broker(p):
    p.terminate()
    try:
        _, err = p.communicate(timeout=60)
    except subprocess.TimeoutExpired:
        p.kill()
        _, err = p.communicate()
    return err


def run_point(pool, mode, nproc, seconds, answers=None, broker_name=None, timeout=300):
    d = tempfile.mkdtemp(prefix="impw_")
    how = "direct" if mode == "direct" else "broker:%s" % broker_name
    cmd = lambda i: [HARNESS, pool, str(seconds), str(i), d, how] + ([answers] if answers else [])
    procs = [subprocess.Popen(cmd(i), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(nproc)]
    t_end = time.time() + timeout
    try:
        while sum(os.path.exists(os.path.join(d, "ready.%d" % i)) for i in range(nproc)) < nproc:
            dead = [p for p in procs if p.poll() is not None]
            if dead or time.time() > t_end:
                raise SystemExit("worker did not get ready: %r" % [p.stderr.read()[-400:] for p in dead])
            time.sleep(0.01)
        open(os.path.join(d, "go"), "w").close()
        out = []
        for p in procs:
            so, se = p.communicate(timeout=timeout)
            if p.returncode != 0:
                raise SystemExit("worker failed (%d): %s" % (p.returncode, se[-800:]))
            out.append(json.loads(so.strip().splitlines()[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(d, ignore_errors=True)
    total = sum(r["requests"] for r in out)
    span = max(r["seconds"] for r in out)
    lat = sorted(out, key=lambda r: r["p50_us"])
    return {
        "mode": mode, "processes": nproc, "requests": total, "seconds": round(span, 3), "requests_per_s": round(total / span, 1),
        "p50_us": round(sum(r["p50_us"] * r["requests"] for r in out) / max(total, 1), 1),
        "p95_us": round(max(r["p95_us"] for r in out), 1), "p99_us": round(max(r["p99_us"] for r in out), 1),
        "mean_batch": round(sum(r["mean_batch"] * r["requests"] for r in out) / max(total, 1), 2),
        "mismatches": sum(r["mismatches"] for r in out), "checked": all(r["checked"] for r in out),
        "chain_timeouts": sum(r["chain_timeouts"] for r in out), "refused": sum(r["refused"] for r in out),
        "slowest_worker_p50_us": lat[-1]["p50_us"], "fastest_worker_p50_us": lat[0]["p50_us"],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["direct", "broker"])
    ap.add_argument("procs", type=int, nargs="+")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--gather-us", type=int, default=0)
    ap.add_argument("--pool", default=os.path.join(ROOT, "gpurun_out", "jpeg_pool.bin"))
    ap.add_argument("--answers", default=None)
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for the broker (0: the runtime's default, 4)")
    ap.add_argument("--split-kb", type=int, default=0, help="a launch takes files up to so many KB, or above (A/B)")
    ap.add_argument("--pipeline", type=int, default=0, help="0: the broker's lanes take one batch at a time (A/B)")
    ap.add_argument("--cu-split", type=int, default=0, help="IMPGPU_LANE_CU_SPLIT for the broker: every lane on its own n-th of the CUs")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.pool), exist_ok=True)
    make_pool(args.pool)
    for n in args.procs:
        if args.mode == "direct" and n > 6:
            print(json.dumps({"mode": "direct", "processes": n, "skipped": "more than 6 processes on the card"}), flush=True)
            continue
        broker = None
        name = "/impgpu-scaling-%d" % os.getpid()
        if args.mode == "broker":
            env = {}
            if args.hw_queues:
                env["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)
            if args.cu_split:
                env["IMPGPU_LANE_CU_SPLIT"] = str(args.cu_split)
            broker = start_broker(name, args.threads, args.gather_us, env=env or None, extra=["--pipeline", str(args.pipeline), "--split-kb", str(args.split_kb)])
        try:
            r = run_point(args.pool, args.mode, n, args.seconds, args.answers, name)
            if broker:
                r["broker_threads"] = args.threads
                r["gather_us"] = args.gather_us
                r["hw_queues"] = args.hw_queues or 8
                r["cu_split"] = args.cu_split
                r["pipeline"] = args.pipeline
            print(json.dumps(r), flush=True)
        finally:
            if broker:
                err = stop_broker(broker)
                for ln in err.strip().splitlines()[-2:]:
                    print("# " + ln, flush=True)


if __name__ == "__main__":
    main()
