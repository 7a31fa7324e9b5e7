"""N > 1 path on CPU: two gloo ranks shard a request list round-robin (no data-path collective), and the only
collectives are the barrier and the max-over-ranks of the elapsed time that bench.py uses."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_lib as orc
    from conftest import noise_image
    from ngx_http_imgproc_amd.shard import round_robin, elapsed_max

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = round_robin(n_items, rank, world)
    # each rank processes only its own frames (the oracle stands in for the device on this CPU-only test)
    digests = {i: int(orc.cv_resize(noise_image(40, 60, 4, i), 14, 10, orc.INTER_CUBIC).astype(np.uint64).sum()) for i in mine}
    dist.barrier()
    t = elapsed_max(0.1 * (rank + 1), dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, digests))
    if rank == 0:
        q.put((t, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_round_robin_two_ranks_gloo():
    world, n_items = 2, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    t, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(t - 0.2) < 1e-9                                   # max over ranks
    owned = [i for mine, _ in gathered for i in mine]
    assert sorted(owned) == list(range(n_items))                 # a partition: every frame exactly once
    assert gathered[0][0] == [0, 2, 4, 6, 8] and gathered[1][0] == [1, 3, 5, 7]
    # sharded result == unsharded result, frame by frame
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    from conftest import noise_image

    merged = {}
    for _, d in gathered:
        merged.update(d)
    for i in range(n_items):
        assert merged[i] == int(orc.cv_resize(noise_image(40, 60, 4, i), 14, 10, orc.INTER_CUBIC).astype(np.uint64).sum())


def test_round_robin_edges():
    from ngx_http_imgproc_amd.shard import round_robin

    assert round_robin(0, 0, 4) == [] and round_robin(3, 3, 4) == [] and round_robin(5, 1, 1 + 1) == [1, 3]
    for world in (1, 2, 4, 8):
        allidx = sorted(i for r in range(world) for i in round_robin(1024, r, world))
        assert allidx == list(range(1024))


def _totals_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ngx_http_imgproc_amd.shard import job_totals, round_robin

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # a stubbed step: each rank "processes" its round-robin share of 1000 requests; rank 1 is the slow one
    mine = round_robin(1000, rank, world)
    seconds = 0.5 if rank == 0 else 0.8
    dist.barrier()
    secs, (requests, src_bytes) = job_totals(seconds, [len(mine), 1000 * len(mine)], dist)
    if rank == 0:
        q.put((secs, requests, src_bytes, requests / secs))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_aggregation_two_ranks_gloo():
    """The function bench.py builds every N > 1 line from (ngx_http_imgproc_amd.shard.job_totals), under two gloo ranks:
    the job's time is the SLOWEST rank's, its units are ALL ranks' -- value = whole-job throughput, as the driver expects."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_totals_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    secs, requests, src_bytes, value = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(secs - 0.8) < 1e-12 and requests == 1000 and src_bytes == 1000 * 1000
    assert abs(value - 1250.0) < 1e-9
    from ngx_http_imgproc_amd.shard import job_totals

    assert job_totals(0.25, [10, 20]) == (0.25, [10.0, 20.0])           # one rank: identity
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("job_totals(") >= 4 and "all_reduce" not in src    # every reporting path of bench.py goes through it
