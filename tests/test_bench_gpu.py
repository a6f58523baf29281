"""bench.py's multi-rank path on a real device: two ranks on the one GPU of the test box, gloo standing in for RCCL (two RCCL
ranks cannot share a device).  Everything but the transport is the code the driver's `--gpus N` runs execute: rank-local
pipelines with `gather=True` (the per-step all-gather of detection records), barrier-bracketed timing, MAX over ranks, the
checked all-gather behind `rccl_ranks`, rank 0's batch-1 latency loop while the others wait at the closing barrier."""
import json
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.gpu


def test_bench_two_ranks_on_one_gpu_over_gloo(device):
    env = dict(os.environ, CP_BENCH_BACKEND="gloo", CP_BENCH_PORT=str(32000 + os.getpid() % 2000))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-legs", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["config"]["global_batch"] == 128
    assert "all-gather" in d["config"]["parallelism"] and d["value"] > 0 and d["roofline"]["kernel"]
    assert d["p50_frame_ms_batch1"] and d["legs"] is None
    assert "skipped" in d["cpu_baseline"]   # N > 1: a stub, so that a SCALE parser never finds the key missing


def _bench(argv, env_extra, launcher=None):
    env = dict(os.environ, CP_BENCH_PORT=str(33000 + os.getpid() % 2000))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra)
    cmd = (launcher or [sys.executable]) + [os.path.join(REPO, "bench.py")] + argv
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="pre-flight for a multi-GPU box: needs two visible GPUs")
def test_bench_two_gpus_over_rccl_preflight(device):
    """The first thing to run on a box with more than one GPU (the builder never had one): `bench.py --gpus 2` exactly as the
    driver's scaling runs launch it, backend "nccl" = RCCL over xGMI -- process-group bring-up with dmabuf IPC
    (HSA_ENABLE_IPC_MODE_LEGACY=0), one device per LOCAL_RANK, the per-step all-gather of detection records, the checked gather
    behind `rccl_ranks`."""
    d = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-legs", "--no-cpu-baseline", "--no-latency"],
               {"CP_BENCH_BACKEND": "nccl", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["config"]["global_batch"] == 128
    assert "RCCL" in d["config"]["parallelism"] and d["value"] > 0 and d["scaling"] == "weak"


def test_bench_single_rank_under_the_launcher_equals_the_plain_launch(device):
    """N = 1 as the driver starts a scaling series (`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`)
    must be the same measurement as `python bench.py`: same workload, batch, no collective, a roofline, and a throughput within
    run-to-run noise of the plain launch."""
    argv = ["--gpus", "1", "--steps", "6", "--warmup", "2", "--no-legs", "--no-cpu-baseline", "--no-latency"]
    plain = _bench(argv, {})
    port = str(34000 + os.getpid() % 2000)
    under = _bench(argv, {}, launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                                       "--master-addr", "127.0.0.1", "--master-port", port])
    for d in (plain, under):
        assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["config"]["global_batch"] == 64
        assert "single rank" in d["config"]["parallelism"] and d["roofline"]["kernel"]
    assert plain["config"] == under["config"] and plain["metric"] == under["metric"]
    assert abs(plain["value"] - under["value"]) / plain["value"] < 0.08
