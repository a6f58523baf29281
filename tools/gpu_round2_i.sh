#!/bin/bash
set -u
O=$PWD/gpurun_out/r2i
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "halo or dcn or backbone_vs_reference or spot_parity" 2>&1 | tail -3
for d in 0 16384 0 16384; do
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-configs2 --no-latency --dbg $d > $O/bench$d.json 2>$O/bench$d.err
python - <<PY
import json
d=json.loads(open("$O/bench$d.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("dbg $d: %.1f img/s  %.3f ms/step roles %s" % (d["value"], d["ms_per_step"], r["ms_per_step_by_role"]))
print({k:(v["tflops"],v["ms_per_step"],v["launches_per_step"]) for k,v in r["all_conv_kernels"].items() if "halo" in k or "n32" in k})
PY
done
