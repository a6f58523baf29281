#!/bin/bash
set -u
O=$PWD/gpurun_out/r2h
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -5 $O/pytest.txt
python tools/dcn_bench.py
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/bench.json 2>$O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("%.1f img/s  %.3f ms/step  p50 B=1 %.3f ms  roles %s" % (d["value"], d["ms_per_step"], d["p50_frame_ms_batch1"], r["ms_per_step_by_role"]))
print({k:(v["tflops"],v["ms_per_step"],v["launches_per_step"]) for k,v in r["all_conv_kernels"].items()})
c=d["configs2"]; print("configs2 %.1f img/s %.3f ms/step roles %s" % (c["value"], c["ms_per_step"], c["ms_per_step_by_role"]))
PY
