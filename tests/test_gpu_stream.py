"""BASELINE configs[4] on the device: a mixed-size request stream (256 px - 4K) driven from several host threads, each
with its own lane (HIP stream, pool, table cache), every output compared with the oracle.  This is the shape of the
reference's per-frame Resize loop on whatever sizes arrive (bridge.c:588-604) under `worker_processes N`.

Also here: the per-lane resize-table cache under eviction pressure (more geometries than it holds, from several
threads at once) and the batch entry points on a caller's stream with no host wait in between.
"""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import oracle_lib as orc
from conftest import noise_image
from ngx_http_imgproc_amd.workloads import MIXED_RESIZE, mixed_sizes

pytestmark = pytest.mark.gpu


def run_threads(n_threads, items, work):
    """Deal `items` to n_threads workers; returns the list of (item, failure text)."""
    failures = []
    lock = threading.Lock()
    it = iter(list(enumerate(items)))

    def worker():
        while True:
            with lock:
                nxt = next(it, None)
            if nxt is None:
                return
            try:
                msg = work(*nxt)
            except Exception as exc:  # noqa: BLE001  (reported through the assertion below)
                msg = repr(exc)
            if msg:
                failures.append((nxt[1], msg))

    ts = [threading.Thread(target=worker) for _ in range(n_threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return failures


def test_mixed_size_request_stream_matches_oracle(gpu):
    """64 seeded requests of bench.py --stream's generator, with one 256 px and one 3840x2160 frame forced in,
    4 host threads, resize=224,0: upload -> Resize() -> download, bit-exact against the oracle's Resize()."""
    sizes = mixed_sizes(64)
    sizes[5] = (256, 144)
    sizes[17] = (3840, 2160)
    sizes[40] = (2160, 3840)
    cfg = gpu.Config()

    def one(i, wh):
        w, h = wh
        rng = np.random.Generator(np.random.PCG64(0x1A4D5000 + i))
        frame = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        im = gpu.Image(frame)
        rc = gpu.lib.impgpu_resize(C.byref(im.h), MIXED_RESIZE, C.byref(cfg.c), 0)
        rc_o, want = orc.resize(frame, MIXED_RESIZE.decode())
        if rc != rc_o:
            return "codes %d / %d" % (rc, rc_o)
        got = im.numpy()
        im.release()
        if got.shape != want.shape or not np.array_equal(got, want):
            return "pixels differ (%r vs %r)" % (got.shape, want.shape)
        return None

    assert run_threads(4, sizes, one) == []


def test_mixed_sizes_through_pinned_uploads_keep_several_requests_in_flight(gpu):
    """The bench loop's own shape: pinned source, `inflight` requests enqueued per thread before one impgpu_sync."""
    sizes = mixed_sizes(48, seed=0x1A4D0A05)
    lib = gpu.lib
    maxpx = max(w * h for w, h in sizes)
    rng = np.random.Generator(np.random.PCG64(77))
    noise = rng.integers(0, 256, size=maxpx * 4, dtype=np.uint8)
    hsrc = lib.impgpu_host_alloc(noise.nbytes)
    C.memmove(hsrc, noise.ctypes.data, noise.nbytes)
    cfg = gpu.Config()
    out_bytes = 224 * 224 * 4 * 4
    groups = [sizes[i:i + 4] for i in range(0, len(sizes), 4)]

    def one(_, group):
        hdst = lib.impgpu_host_alloc(out_bytes * len(group))
        live, shapes = [], []
        for k, (w, h) in enumerate(group):
            img = C.c_void_p()
            assert lib.impgpu_image_upload_pinned(hsrc, w, h, 4, w * 4, C.byref(img)) == 0
            assert lib.impgpu_resize(C.byref(img), MIXED_RESIZE, C.byref(cfg.c), 0) == 0
            ow, oh = lib.impgpu_image_width(img), lib.impgpu_image_height(img)
            assert lib.impgpu_image_download_pinned(img, hdst + out_bytes * k, ow * 4) == 0
            live.append(img)
            shapes.append((oh, ow))
        assert lib.impgpu_sync() == 0
        bad = None
        for k, (w, h) in enumerate(group):
            oh, ow = shapes[k]
            got = np.ctypeslib.as_array((C.c_uint8 * (oh * ow * 4)).from_address(hdst + out_bytes * k)).reshape(oh, ow, 4)
            rc_o, want = orc.resize(noise[:w * h * 4].reshape(h, w, 4), MIXED_RESIZE.decode())
            if rc_o != 0 or not np.array_equal(got, want):
                bad = "request %dx%d differs" % (w, h)
        for img in live:
            lib.impgpu_image_release(C.byref(img))
        lib.impgpu_host_free(hdst)
        return bad

    failures = run_threads(3, groups, one)
    lib.impgpu_host_free(hsrc)
    assert failures == []


def test_table_cache_eviction_under_threads(gpu):
    """More distinct geometries than a lane's table cache holds (256), from three threads at once: evicted tables go
    back to the pool in stream order, so no launch may ever read a recycled table (round 1's cache freed tables that a
    kernel about to be enqueued still pointed at)."""
    geoms = [(40 + (i % 37), 30 + (i // 37) % 29, 9 + i % 7) for i in range(700)]     # (w, h, target w): 700 distinct keys
    assert len(set(geoms)) == 700
    cfg = gpu.Config()

    def one(i, g):
        w, h, tw = g
        frame = noise_image(h, w, 4 if i % 2 else 3, 3000 + i)
        modes = [("%d,0" % tw, 0), ("%d,0,up" % (w + tw), 0), ("%d,0" % tw, 1)]     # AREA, CUBIC (enlarging), NN
        args, simple = modes[i % 3]
        im = gpu.Image(frame)
        rc = gpu.lib.impgpu_resize(C.byref(im.h), args.encode(), C.byref(cfg.c), simple)
        rc_o, want = orc.resize(frame, args, simple=simple)
        got = im.numpy()
        im.release()
        return None if (rc == rc_o == 0 and np.array_equal(got, want)) else "geometry %r differs" % (g,)

    assert run_threads(3, geoms, one) == []


def test_batch_calls_on_a_callers_stream_need_no_host_wait(gpu):
    """The batch entry points take a hipStream_t and return after enqueue (impgpu.h): temporary blocks are parked
    behind an event of that stream instead of a hipStreamSynchronize.  Many back-to-back calls with different
    geometries and LUTs, one sync at the end, everything compared with the oracle."""
    import torch

    stream = torch.cuda.Stream()
    n = 6
    src_np = np.stack([noise_image(96, 128, 4, 4100 + i) for i in range(n)])
    src = torch.from_numpy(src_np).cuda()
    cfg = gpu.Config(allow_experiments=True)
    ov = noise_image(8, 12, 4, 4200)
    assert cfg.prepare_watermark(ov, "r", "b", 2, 2, 60) == 0
    outs = []
    torch.cuda.synchronize()
    for k, (rw, rh) in enumerate([(64, 48), (50, 40), (64, 48), (33, 21), (70, 60), (50, 40)]):
        dst = torch.zeros((n, rw, rh, 4), dtype=torch.uint8, device="cuda")       # rotated: rh wide, rw tall
        gpu.batch_resize_rotate_watermark(src.data_ptr(), 96 * 128 * 4, 128, 96, 128 * 4, dst.data_ptr(), rw * rh * 4, rh * 4,
                                          rw, rh, 90, cfg, 4, n, stream=stream.cuda_stream)
        fl = torch.from_numpy(src_np.copy()).cuda()
        torch.cuda.current_stream().synchronize()
        stream.wait_stream(torch.cuda.current_stream())
        assert gpu.batch_filters(fl.data_ptr(), 96 * 128 * 4, 128, 96, 4, 128 * 4, n, ["gamma=%s" % (1.2 + 0.1 * k), "kelvin=1"],
                                 stream=stream.cuda_stream) == 0
        outs.append(((rw, rh), dst, fl, 1.2 + 0.1 * k))
    stream.synchronize()
    for (rw, rh), dst, fl, gamma in outs:
        got, gotf = dst.cpu().numpy(), fl.cpu().numpy()
        for i in range(n):
            rc, o = orc.resize(src_np[i], "%d,%d" % (rw, rh))
            rc, o = orc.filter(o, "rotate=90")
            rc, o = orc.watermark(o, ov, "r", "b", 2, 2, 60)
            assert np.array_equal(got[i], o), (rw, rh, i)
            rc, f = orc.filter(src_np[i], "gamma=%s" % gamma)
            rc, f = orc.filter(f, "kelvin=1", 1)
            assert np.array_equal(gotf[i], f), (gamma, i)
    cfg.release()


def _mixed_batch(gpu, frames_np, targets, simple=False, stream=None):
    """Upload every frame to its own tensor, run impgpu_batch_resize_mixed once, return (rc, outputs)."""
    import torch

    c = frames_np[0].shape[2]
    srcs = [torch.from_numpy(f).cuda() for f in frames_np]
    dsts = [torch.zeros((dh, dw, c), dtype=torch.uint8, device="cuda") for dw, dh in targets]
    torch.cuda.synchronize()
    items = [(s.data_ptr(), f.shape[1], f.shape[0], f.shape[1] * c, d.data_ptr(), dw, dh, dw * c)
             for s, f, d, (dw, dh) in zip(srcs, frames_np, dsts, targets)]
    rc = gpu.batch_resize_mixed(items, c, simple=simple, stream=stream)
    if stream is None:
        gpu.sync()
    return rc, dsts


@pytest.mark.parametrize("c", [3, 4])
def test_mixed_geometry_batch_is_the_per_frame_resize(gpu, c):
    """impgpu_batch_resize_mixed: BASELINE configs[4] on resident frames.  40 sizes of the stream generator plus the
    geometries that must NOT ride the descriptor launch -- exact 2x / 4x / 3x factors (resizeAreaFast_), an enlargement
    (CUBIC), runs longer than 16 pixels (a 3840-wide frame to 100), a 1x1 frame -- every output compared with the
    oracle's cvResize under the interpolation Resize() picks (bridge.c:188-192)."""
    sizes = mixed_sizes(40, seed=0x1A4D0A07)
    sizes = [(min(w, 1400), min(h, 1400)) for w, h in sizes]                  # keep the oracle's share of the run short
    sizes += [(448, 300), (896, 448), (672, 99), (100, 60), (3840, 64), (1, 1), (224, 224), (225, 224)]
    cfg = gpu.Config()
    frames, targets = [], []
    for i, (w, h) in enumerate(sizes):
        frames.append(noise_image(h, w, c, 5200 + i))
        if (w, h) == (3840, 64):
            targets.append((100, 5))
        elif (w, h) == (448, 300):
            targets.append((224, 150))                                         # exact 2x2
        elif (w, h) == (896, 448):
            targets.append((224, 112))                                         # exact 4x4
        elif (w, h) == (672, 99):
            targets.append((224, 33))                                          # exact 3x3
        else:
            rc, (dw, dh, _) = gpu.resize_geometry(w, h, "224,0,up", cfg)
            assert rc == 0
            targets.append((dw, dh))
    rc, outs = _mixed_batch(gpu, frames, targets)
    assert rc == 0
    for f, (dw, dh), out in zip(frames, targets, outs):
        interp = orc.INTER_CUBIC if (dw > f.shape[1] or dh > f.shape[0]) else orc.INTER_AREA
        want = orc.cv_resize(f, dw, dh, interp)
        assert np.array_equal(out.cpu().numpy(), want), (f.shape, dw, dh)


def test_mixed_geometry_batch_outlives_the_table_cache(gpu):
    """600 distinct geometries in one call: the launcher must flush its descriptor launches before the lane's
    256-entry table cache recycles a table they point at; then the same call on a caller's stream, and with simple=1 (NN)."""
    import torch

    geoms = [(60 + (i % 41), 40 + (i // 41) % 23, 11 + i % 13) for i in range(600)]
    assert len(set(geoms)) == 600
    frames = [noise_image(h, w, 4, 6000 + i) for i, (w, h, _) in enumerate(geoms)]
    targets = [(tw, max(1, round(h * tw / w))) for (w, h, tw) in geoms]
    rc, outs = _mixed_batch(gpu, frames, targets)
    assert rc == 0
    for f, (dw, dh), out in zip(frames, targets, outs):
        assert np.array_equal(out.cpu().numpy(), orc.cv_resize(f, dw, dh, orc.INTER_AREA)), (f.shape, dw, dh)
    stream = torch.cuda.Stream()
    rc, outs = _mixed_batch(gpu, frames[:150], targets[:150], stream=stream.cuda_stream)
    assert rc == 0
    stream.synchronize()
    for f, (dw, dh), out in zip(frames[:150], targets[:150], outs):
        assert np.array_equal(out.cpu().numpy(), orc.cv_resize(f, dw, dh, orc.INTER_AREA)), (f.shape, dw, dh)
    rc, outs = _mixed_batch(gpu, frames[:40], targets[:40], simple=True)
    assert rc == 0
    for f, (dw, dh), out in zip(frames[:40], targets[:40], outs):
        assert np.array_equal(out.cpu().numpy(), orc.cv_resize(f, dw, dh, orc.INTER_NN)), (f.shape, dw, dh)


def test_mixed_geometry_batch_rejects_a_malformed_item_before_launching(gpu):
    import torch

    frames = [noise_image(50, 70, 4, 6900 + i) for i in range(3)]
    srcs = [torch.from_numpy(f).cuda() for f in frames]
    dsts = [torch.full((20, 30, 4), 7, dtype=torch.uint8, device="cuda") for _ in frames]
    items = [(s.data_ptr(), 70, 50, 280, d.data_ptr(), 30, 20, 120) for s, d in zip(srcs, dsts)]
    items[2] = items[2][:7] + (100,)                                           # destination pitch shorter than a row
    assert gpu.batch_resize_mixed(items, 4) == gpu.IMP_ERROR_INVALID_ARGS
    gpu.sync()
    assert all(int(d.min()) == 7 and int(d.max()) == 7 for d in dsts)          # the two good items were not run either
    assert gpu.batch_resize_mixed([], 4) == 0


@pytest.mark.parametrize("c", [3, 4])
def test_pinned_transfers_with_foreign_row_pitches(gpu, c):
    """impgpu_image_upload_pinned / _download_pinned with host pitches other than the frame's own: tightly packed BGR rows
    (what libjpeg hands out), rows padded far beyond the frame's; one linear DMA + a device re-pitch each way."""
    lib = gpu.lib
    w, h = 203, 57
    frame = noise_image(h, w, c, 7300 + c)
    for pitch in (w * c, w * c + 13, ((w * c + 3) & ~3)):
        hsrc = lib.impgpu_host_alloc(pitch * h)
        buf = np.ctypeslib.as_array((C.c_uint8 * (pitch * h)).from_address(hsrc)).reshape(h, pitch)
        buf[:] = 0xEE
        buf[:, :w * c] = frame.reshape(h, w * c)
        img = C.c_void_p()
        assert lib.impgpu_image_upload_pinned(hsrc, w, h, c, pitch, C.byref(img)) == 0
        assert lib.impgpu_filter(C.byref(img), b"flip=10", 1) == 0
        opitch = w * c + 5
        hdst = lib.impgpu_host_alloc(opitch * h)
        out = np.ctypeslib.as_array((C.c_uint8 * (opitch * h)).from_address(hdst)).reshape(h, opitch)
        out[:] = 0x77
        assert lib.impgpu_image_download_pinned(img, hdst, opitch) == 0
        assert lib.impgpu_sync() == 0
        assert np.array_equal(out[:, :w * c].reshape(h, w, c), frame[:, ::-1]), pitch
        assert not out[:-1, w * c:].any()                   # the gaps between rows arrive as zeros
        lib.impgpu_image_release(C.byref(img))
        lib.impgpu_host_free(hsrc)
        lib.impgpu_host_free(hdst)


@pytest.mark.gpu
def test_pool_gives_memory_back_past_its_cap(tmp_path):
    """A lane's free list is trimmed above IMPGPU_POOL_CAP_MB when its stream has been waited for: a worker that once saw
    large frames does not keep their buckets for ever (N workers x lanes share one GPU)."""
    import subprocess
    import sys

    script = r'''
import numpy as np
import torch  # first (see tests/conftest.py)
import ngx_http_imgproc_amd as imp
imp.env_start(0)
torch.cuda.init()
free0 = torch.cuda.mem_get_info()[0]
ims = [imp.Image(np.zeros((2160, 3840, 4), dtype=np.uint8)) for _ in range(6)]      # 6 x 33 MB
imp.sync()
held = free0 - torch.cuda.mem_get_info()[0]
for im in ims:
    im.release()
imp.sync()                                           # the wait after which the free list is looked at
kept = free0 - torch.cuda.mem_get_info()[0]
small = imp.Image(np.zeros((64, 64, 4), dtype=np.uint8))   # the pool still works afterwards
ok = small.numpy().shape == (64, 64, 4)
imp.env_destroy()
print(held, kept, ok)
'''
    import os
    from conftest import ROOT
    for cap, limit in (("64", 80 << 20), ("0", None)):
        p = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, cwd=ROOT,
                           env=dict(os.environ, IMPGPU_POOL_CAP_MB=cap, PYTHONPATH=ROOT))
        assert p.returncode == 0, p.stderr[-2000:]
        held, kept, ok = p.stdout.strip().split("\n")[-1].split()
        assert ok == "True" and int(held) >= 6 * 33000000
        if limit is not None:
            assert int(kept) < limit, (held, kept)           # trimmed to half the cap or less
        else:
            assert int(kept) >= 6 * 33000000                 # cap 0: never trimmed


@pytest.mark.gpu
def test_numa_node_of_the_device_and_thread_binding(gpu):
    """the env knows the NUMA node its device hangs off; impgpu_env_bind_thread puts the calling thread on that
    node's CPUs (or says it cannot: one-node host, foreign cpuset).  Run in a thread so the test process keeps its mask."""
    import os

    node = gpu.lib.impgpu_env_numa_node()
    assert node >= -1
    result = {}

    def work():
        before = os.sched_getaffinity(0)
        rc = gpu.lib.impgpu_env_bind_thread()
        result["rc"], result["before"], result["after"] = rc, before, os.sched_getaffinity(0)

    t = threading.Thread(target=work)
    t.start()
    t.join()
    if result["rc"] == 0:
        assert node >= 0 and result["after"] and result["after"] <= result["before"]
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        assert result["after"] <= cpus
    else:
        assert result["rc"] == 1 and result["after"] == result["before"]      # IMP_ERROR_UNSUPPORTED: nothing changed


def _where_is_it_stuck(pid):
    """What the kernel says about every thread of a child that did not come back: name, state, wait channel, system call,
    kernel stack where readable.  Goes into the failure text, so that a hang names its own location."""
    import glob

    lines = []
    for task in sorted(glob.glob("/proc/%d/task/*" % pid)):
        row = [os.path.basename(task)]
        for item in ("comm", "wchan", "syscall", "stack"):
            try:
                with open(os.path.join(task, item)) as f:
                    row.append("%s=%s" % (item, " | ".join(f.read().split("\n")[:12]).strip()))
            except OSError as e:
                row.append("%s=<%s>" % (item, e.strerror))
        try:
            with open(os.path.join(task, "status")) as f:
                row.append([ln.strip() for ln in f if ln.startswith("State:")][0])
        except (OSError, IndexError):
            pass
        lines.append("  ".join(row))
    return "\n".join(lines)


def _run_child(code, timeout=240):
    """Run `code` in a fresh interpreter; (returncode, stdout, stderr), or on a timeout an AssertionError that carries
    the steps the child had printed and where each of its threads was."""
    import subprocess
    import sys
    import tempfile

    with tempfile.TemporaryFile("w+") as so, tempfile.TemporaryFile("w+") as se:
        p = subprocess.Popen([sys.executable, "-u", "-c", code], stdout=so, stderr=se, text=True)
        try:
            p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            where = _where_is_it_stuck(p.pid)
            p.kill()
            p.wait()
            so.seek(0)
            se.seek(0)
            raise AssertionError("child hung after %d s\nsteps printed: %s\nstderr: %s\nthreads:\n%s"
                                 % (timeout, so.read()[-2000:], se.read()[-2000:], where))
        so.seek(0)
        se.seek(0)
        return p.returncode, so.read(), se.read()


def test_import_order_with_torch_does_not_matter():
    """libimpgpu.so before torch used to leave torch without a device (two HIP runtimes in one process); _lib.py now puts
    both on torch's copy.  Checked in a child process, because this one imported torch first (conftest).  The child ends
    the way a script ends -- no env_destroy, an ordinary interpreter exit with the env alive and torch on the same runtime:
    impgpu_env_start's atexit hook gives the env back before the runtime's own handlers run (round 4 had one child that
    never came back from here, with nothing on record to say where; every step now prints a line as it completes, and a
    timeout reports every thread's wait channel)."""
    code = (
        "import faulthandler, sys, numpy as np\n"
        "faulthandler.dump_traceback_later(200, exit=True)\n"          # a hang inside a call says where, on stderr
        "sys.path.insert(0, %r)\n"
        "def step(s): print(s, flush=True)\n"
        "import ngx_http_imgproc_amd as gpu; step('1 library loaded')\n"
        "assert 'torch' not in sys.modules\n"
        "gpu.env_start(0); step('2 env started')\n"
        "a = np.arange(64 * 64 * 4, dtype=np.uint8).reshape(64, 64, 4)\n"
        "im = gpu.Image(a); assert im.cv_resize(32, 32, gpu.INTER_AREA) == 0; out = im.numpy(); step('3 first resize')\n"
        "import torch; step('4 torch imported')\n"
        "assert torch.cuda.is_available(), 'torch lost the device'\n"
        "t = torch.arange(1024, device='cuda').sum().item(); assert t == 1023 * 512; step('5 torch kernel')\n"
        "im2 = gpu.Image(a); assert im2.cv_resize(32, 32, gpu.INTER_AREA) == 0\n"
        "assert np.array_equal(im2.numpy(), out); step('6 second resize')\n"
        "step('ok, leaving with the env alive')\n"
    ) % str(__import__("pathlib").Path(__file__).resolve().parent.parent)
    rc, so, se = _run_child(code)
    assert rc == 0 and "ok, leaving" in so, (rc, so, se[-2000:])


def test_exit_with_a_live_env_from_other_threads_too():
    """A worker that is told to quit between requests: lanes of two threads alive (streams, blocking-sync events, pinned
    rings, pool blocks, a prepared watermark), no impgpu_env_destroy, plain exit().  The atexit hook registered by
    impgpu_env_start tears them down; the process must end by itself and with status 0."""
    code = (
        "import sys, threading, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "def step(s): print(s, flush=True)\n"
        "import ngx_http_imgproc_amd as gpu\n"
        "gpu.env_start(0); step('env')\n"
        "a = np.arange(256 * 256 * 3, dtype=np.uint8).reshape(256, 256, 3)\n"
        "def work():\n"
        "    im = gpu.Image(a); assert im.cv_resize(64, 64, gpu.INTER_CUBIC) == 0; im.numpy()\n"
        "ts = [threading.Thread(target=work) for _ in range(2)]\n"
        "[t.start() for t in ts]; [t.join() for t in ts]; step('threads done')\n"
        "keep = gpu.Image(a); assert keep.cv_resize(128, 128, gpu.INTER_AREA) == 0; step('frame kept alive')\n"
        "step('leaving')\n"
        "sys.exit(0)\n"
    ) % str(__import__("pathlib").Path(__file__).resolve().parent.parent)
    rc, so, se = _run_child(code, timeout=120)
    assert rc == 0 and "leaving" in so, (rc, so, se[-2000:])
    assert "still busy at exit" not in se
