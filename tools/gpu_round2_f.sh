#!/bin/bash
set -u
O=$PWD/gpurun_out/r2f
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -6 $O/pytest.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("%.1f img/s  %.3f ms/step  p50 B=1 %.3f ms  roles %s" % (d["value"], d["ms_per_step"], d["p50_frame_ms_batch1"], r["ms_per_step_by_role"]))
print("dcn", r["dcn"]); print("decode", r["decode"])
c=d["configs2"]; print("configs2 %.1f img/s %.3f ms/step roles %s" % (c["value"], c["ms_per_step"], c["ms_per_step_by_role"]))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["faithful"]["value"])
PY
