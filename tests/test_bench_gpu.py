"""bench.py's multi-rank path on a real device: two ranks on the one GPU of the test box, gloo standing in for RCCL (two RCCL
ranks cannot share a device).  Everything but the transport is the code the driver's `--gpus N` runs execute: rank-local
pipelines with `gather=True` (the per-step all-gather of detection records), barrier-bracketed timing, MAX over ranks, the
checked all-gather behind `rccl_ranks`, rank 0's batch-1 latency loop while the others wait at the closing barrier."""
import json
import os
import subprocess
import sys

import pytest

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
