mkdir -p gpurun_out/c4
python -m pytest tests/test_tracking_loop.py tests/test_gpu_pose_chain.py tests/test_gpu_parity.py -m gpu -q --tb=short -k "tracking or tracker or pose_chain or grouped" 2>&1 | grep -v "^Fix size\|^training chunk\|^The output\|^heads \|^Creating\|^loaded" | tail -60 > gpurun_out/c4/pytest.txt
tail -5 gpurun_out/c4/pytest.txt
python bench.py --workload track_e2e --steps 8 --warmup 2 2>gpurun_out/c4/e2e.err | tail -1 > gpurun_out/c4/track_e2e.json
python - <<PY
import json
d=json.load(open("gpurun_out/c4/track_e2e.json"))
print(json.dumps({k:d[k] for k in ("value","ms_per_step","detection_threshold","device_tracker","host_tracker")}, indent=1))
PY
B="--no-cpu-baseline --no-legs"
python bench.py $B 2>/dev/null | tail -1 > gpurun_out/c4/grouped.json
python bench.py $B --dbg 16777216 2>/dev/null | tail -1 > gpurun_out/c4/perhead.json
python - <<PY
import json
for n in ("grouped","perhead"):
    d=json.load(open("gpurun_out/c4/%s.json"%n))
    print(n, d["value"], d["ms_per_step"], d["p50_frame_ms_batch1"], d["roofline"]["achieved"], d["roofline"]["ms_per_step_by_role"])
PY
