"""bench.py's multi-rank plumbing without a device: `python bench.py --gpus 2 --dry-run` must re-execute itself under
torch.distributed.run (127.0.0.1), bring up a world-size-2 process group ($CP_BENCH_BACKEND=gloo stands in for RCCL),
run the barrier-bracketed timed region with the per-step all-gather of detection records, reduce the time with MAX over
ranks and print ONE JSON line whose n_gpus / rccl_ranks come from the group that actually ran (the
flag used to be parsed and ignored)."""
import json
import os
import subprocess
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(argv, extra_env=None):
    env = dict(os.environ, CP_BENCH_BACKEND="gloo", CP_BENCH_PORT=str(31000 + os.getpid() % 2000))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line expected, got %d:\n%s" % (len(lines), r.stdout[-2000:])
    return json.loads(lines[0])


def test_bench_gpus_flag_spawns_that_many_ranks():
    d = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2
    assert d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"] == 128
    assert "configs[2]" in d["config"]["workload"] and "x2" in d["config"]["parallelism"]
    assert d["value"] > 0 and abs(d["value"] - 128 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3
    assert d["data"].startswith("dry-run")


def test_bench_single_rank_dry_run_has_no_group():
    d = _run(["--dry-run", "--steps", "2", "--warmup", "0"])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["config"]["global_batch"] == 64
