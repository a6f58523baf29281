"""bench.py's multi-rank plumbing without a device: `python bench.py --gpus 2 --dry-run` must re-execute itself under
torch.distributed.run (127.0.0.1), bring up a world-size-2 process group ($CP_BENCH_BACKEND=gloo stands in for RCCL),
run the barrier-bracketed timed region with the per-step all-gather of detection records, reduce the time with MAX over
ranks and print ONE JSON line whose n_gpus / rccl_ranks come from the group that actually ran (the
flag used to be parsed and ignored)."""
import json

import pytest
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
    assert d["data"].startswith("dry-run") and d["dry_run_devices"] == [0, 1]


def test_bench_single_rank_dry_run_has_no_group():
    d = _run(["--dry-run", "--steps", "2", "--warmup", "0"])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["config"]["global_batch"] == 64


def test_bench_eight_ranks_dress_rehearsal():
    """BASELINE configs[3] plumbing without hardware: 8 ranks over gloo, 64 images each, one checked all-gather; the line
    carries a cpu_baseline stub for N > 1 so that a SCALE parser never finds the key missing."""
    d = _run(["--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["config"]["global_batch"] == 512
    assert "x8" in d["config"]["parallelism"] and d["scaling"] == "weak"
    assert d["dry_run_devices"] == list(range(8))   # rank r would bind device LOCAL_RANK = r (one GPU per rank)
    assert isinstance(d["cpu_baseline"], dict) and "skipped" in d["cpu_baseline"]
    assert "roofline" in d and "legs" in d
    # the eight shards (distributed.shard_range of the global batch; frames are seeded by global image index) tile [0, 512)
    # exactly and arrive in rank order -- bench.py itself refuses to print a line otherwise; the line shows first / last / count
    assert d["dry_run_global_images"] == [0, 511, 512]
    # the collective's own time next to the step (here: gloo, host clock; on the GPU box: HIP events on the launch stream)
    ag = d["allgather"]
    assert ag["ms_per_step_max_over_ranks"] >= ag["ms_per_step_rank0"] > 0
    assert ag["bytes_received_per_rank"] == 512 * 100 * 118 * 4 and ag["gbps_per_rank"] > 0


def test_bench_eight_ranks_tracking_workload_gathers_in_rank_image_slot_order():
    """BASELINE configs[4]'s collective (`--workload track`: every rank's tracker needs every image's detections) rehearsed dry:
    bench.py checks the gathered records' (rank, image, slot) tags and exits non-zero if the order is anything else."""
    d = _run(["--gpus", "8", "--dry-run", "--workload", "track", "--steps", "2", "--warmup", "1"])
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["config"]["global_batch"] == 8 * 16
    assert d["dry_run_global_images"] == [0, 127, 128] and d["dry_run_devices"] == list(range(8))
    assert d["allgather"]["bytes_received_per_rank"] == 128 * 100 * 118 * 4


def test_shards_tile_the_global_batch_for_every_world_size():
    """distributed.shard_range (what bench.py seeds and indexes by): contiguous, disjoint, in rank order, covering [0, n) for
    every world size of the scaling series, divisible or not (opts.py:358-367's chunk sizes without the master-GPU case)."""
    from centerpose_amd.distributed import shard_range

    for world in (1, 2, 4, 8):
        for n in (512, 64 * world, 500, 7):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [e - s for s, e in edges]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


@pytest.mark.parametrize("record", ["r03_bench_default.json", "r04_bench_detail.json", "r05_bench_detail.json"])
def test_compact_line_keeps_every_leg_inside_the_drivers_tail(record):
    """The driver stores an 8 KB tail of stdout: the round-3 line was longer and lost four legs.  The compact form of that very
    record (profiles/r03_bench_default.json, a full round-3 line) and of the round-4 full record (the detail file, with the
    decode-only and rendered-heads legs) must fit with every leg's value and roofline fraction."""
    sys.path.insert(0, REPO)
    import bench

    with open(os.path.join(REPO, "profiles", record)) as f:
        text = f.read().strip()
    full = json.loads(text if text.startswith("{\n") else text.splitlines()[-1])   # (the detail file is pretty-printed)
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 8000, len(text)   # (the driver keeps an 8 KB tail; round 5's line with the e2e_u8 / rendered_e2e legs: 4.7 KB)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in line
    assert line["roofline"]["kernel"] and line["roofline"]["frac"] > 0 and "dcn" in line["roofline"]
    assert set(full["legs"]) == set(line["legs"])
    for name, leg in line["legs"].items():
        assert leg["value"] > 0 and leg["ms_per_step"] > 0, name
        if full["legs"][name].get("roofline"):
            assert leg["roofline"]["frac"] > 0, name
    assert line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and line["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]


def test_roofline_traffic_is_attached_only_to_the_profiled_workload():
    """profiles/pmc_traffic.json holds the PMC passes of ONE workload at ONE batch (its _meta); round 4 printed its figure for
    the head kernel under legs of other networks and batches.  Same workload / batch / precision -> the bytes; anything else ->
    null plus a note."""
    sys.path.insert(0, REPO)
    import bench

    with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as f:
        pmc = json.load(f)
    meta = pmc["_meta"]
    name = "halo16_head_f16x3_m128n128"
    assert name in pmc and meta["workload"] == "full" and meta["batch"] == 64
    prof = {name: dict(launches=2, ms=10.0, flops=4.0e12, bytes=8.0e8)}
    r = bench.roofline_object(prof, {}, 2, 64, "f16x3", "full")
    assert r["traffic"] == pmc[name]["hbm_bytes_per_launch"] and "traffic_note" not in r
    for batch, wl, prec in ((16, "full", "f16x3"), (64, "track", "f16x3"), (64, "full", "f32"), (64, None, "f16x3")):
        r = bench.roofline_object(prof, {}, 2, batch, prec, wl)
        assert r["traffic"] is None and "traffic_note" in r, (batch, wl, prec)
