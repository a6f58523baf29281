"""Multi-process CPU tests (gloo, world_size 2) of the batch-shard + detection all-gather path."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from centerpose_amd import distributed as cpd

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 511, 512):
        for world in (1, 2, 3, 8):
            spans = [cpd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from centerpose_amd import distributed as d

    r, w = d.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # even shards: every rank owns 3 images of K=100 records of 118 floats
    n_global = 6
    s, e = d.shard_range(n_global, r, w)
    full = torch.arange(n_global * 100 * 118, dtype=torch.float32).view(n_global, 100, 118)
    got = d.allgather_detections(full[s:e].clone())
    ok1 = torch.equal(got, full)
    # ragged shards: 5 images over 2 ranks -> 3 + 2
    n_global = 5
    s, e = d.shard_range(n_global, r, w)
    full = torch.arange(n_global * 4 * 118, dtype=torch.float32).view(n_global, 4, 118)
    got = d.allgather_detections_ragged(full[s:e].clone())
    ok2 = torch.equal(got, full)
    # compact gather: only records above the score threshold travel, tagged with their global image index
    g = torch.Generator().manual_seed(5)
    full = torch.rand(4, 10, 118, generator=g)
    mine = full[rank * 2:(rank + 1) * 2].clone()
    rec, idx = d.allgather_detections_compact(mine, 0.7)
    keep = full[..., 4] > 0.7
    ok3 = torch.equal(rec, full[keep]) and torch.equal(idx, torch.arange(4).view(4, 1).expand(4, 10)[keep])
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    q.put((rank, ok1, ok2 and ok3))


def test_allgather_detections_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, True), (1, True, True)]
