mkdir -p gpurun_out/c3
python -m pytest tests/test_gpu_pose_chain.py tests/test_tracking_loop.py tests/test_gpu_parity.py -m gpu -q --tb=short -k "pose_chain or tracking or grouped or dcn" 2>&1 | grep -v "^Fix size\|^training chunk\|^The output\|^heads \|^Creating\|^loaded" | tail -40 > gpurun_out/c3/pytest.txt
B="--no-cpu-baseline --no-legs"
python bench.py $B 2>/dev/null | tail -1 > gpurun_out/c3/grouped.json
python bench.py $B --dbg 16777216 2>/dev/null | tail -1 > gpurun_out/c3/perhead.json
python - <<PY
import json
for n in ("grouped","perhead"):
    d=json.load(open("gpurun_out/c3/%s.json"%n))
    print(n, d["value"], d["ms_per_step"], d["p50_frame_ms_batch1"], d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["ms_per_step_by_role"])
PY
