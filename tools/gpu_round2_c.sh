#!/bin/bash
set -u
O=$PWD/gpurun_out/r2c
mkdir -p $O
export TMPDIR=/tmp
for d in 0 512; do
  python tools/lat_probe.py --dbg $d
done
timeout 600 python -m pytest tests -m gpu -q -x -k "range_safe or small_weights or dcn or backbone_vs_reference or fused_head or hourglass" 2>&1 | tail -2
for d in 0 512; do
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-configs2 --dbg $d > $O/bench_dbg$d.json 2>$O/bench_dbg$d.err
  python - <<PY
import json
d=json.loads(open("$O/bench_dbg$d.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("dbg $d: %.1f img/s  %.3f ms/step  p50 B=1 %.3f ms  roles %s" % (d["value"], d["ms_per_step"], d["p50_frame_ms_batch1"], r["ms_per_step_by_role"]))
PY
done
