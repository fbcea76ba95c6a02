"""CPU, world_size 2 over gloo: the N>1 orchestration bench.py uses (token ranges, barrier-bracketed
timing, MAX over ranks).  The data path itself needs no collective."""
import os
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rqae_amd import sharding


def test_shard_ranges_cover_without_overlap():
    for n in (0, 1, 7, 16, 1000003, 4 * 1024 * 1024):
        for world in (1, 2, 3, 8):
            r = [sharding.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [e - s for s, e in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, end = sharding.shard_range(1001, rank, world)
        done = []

        def step():                      # stands for one encode over this rank's token range
            time.sleep(0.01 * (rank + 1))
            done.append(end - start)

        sec = sharding.timed_steps(step, steps=5, warmup=2, barrier=dist.barrier)
        worst = sharding.max_over_ranks(sec, dist)
        counts = torch.tensor([float(sum(done[2:]))])
        dist.all_reduce(counts)         # test-only: total tokens processed in the timed region
        if rank == 0:
            out.put((sec, worst, counts.item(), len(done)))
    finally:
        dist.destroy_process_group()


def test_two_rank_timing_and_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    sec0, worst, tokens, calls = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert calls == 7                               # 2 warm-up + exactly 5 timed steps
    assert tokens == 5 * 1001                       # both shards, every timed step, nothing twice
    assert worst >= 5 * 0.02 * 0.9                  # the slower rank (20 ms steps) sets the time
    assert worst >= sec0 * 0.999
    assert sharding.aggregate_throughput(1000, 2, 5, 0.5) == 20000.0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts one process per rank (it never touches a
    GPU), the ranks rendezvous, time exactly K steps between barriers, and rank 0 prints ONE JSON line.  --stub swaps the
    encode for a sleep and RCCL for gloo so the orchestration runs on CPU."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--stub"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world_size_seen"] == 2 and rec["steps"] == 4
    assert rec["config"]["tokens_per_gpu_per_step"] == 524288          # BASELINE configs[2]: 4M tokens over 8 GPUs
    assert rec["ms_per_step"] >= 4.0 * 0.9                              # the slower rank (4 ms steps) sets the time
    assert abs(rec["value"] - 2 * 524288 * 4 / (rec["ms_per_step"] * 4e-3)) < 1e-3 * rec["value"]
