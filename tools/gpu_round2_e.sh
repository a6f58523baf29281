#!/bin/bash
set -u
O=$PWD/gpurun_out/r2e
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests -m gpu -q -k "pnp or detector or tracking or halo or decode" 2>&1 | tail -6
timeout 300 python bench.py --workload full --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_full.json 2>$O/bench_full.err
python - <<PY
import json
d=json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
print("full: %.1f img/s %.3f ms/step p50 %.3f roles %s" % (d["value"], d["ms_per_step"], d["p50_frame_ms_batch1"], d["roofline"]["ms_per_step_by_role"]))
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --workload full --steps 8 --warmup 2 --no-cpu-baseline --no-latency > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp $f $O/full_kernel_stats.csv; rm -rf $O/kt
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/full_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (10 steps incl warmup): %.2f" % (tot/10/1e6))
for r in rows:
    if any(k in r["Name"] for k in ("pnp","peaks","assoc","postprocess","at::","elementwise")): print("%-90s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
