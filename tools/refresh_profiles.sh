#!/bin/bash
# Regenerate the judged artefacts under gpurun_out/refresh/ on the GPU box (copy into profiles/ afterwards):
#   the default bench line (BASELINE configs[2] on top, every other configuration as a nested leg), the per-launch layer
#   CSV, rocprofv3 kernel-trace stats and the two PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, no other
#   tracing) of the default bench command's timed region.
#   usage: tools/refresh_profiles.sh [quick]     quick = bench line + kernel stats only
set -u
R=$PWD
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
B="--no-cpu-baseline --no-latency --no-legs"
CP_PROFILE_DUMP=$O/layers_default.csv timeout 300 python bench.py --steps 8 --warmup 2 $B 2>/dev/null | tail -1 > $O/bench_layers.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 10 --warmup 3 $B > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/rocprof_kernel_stats.csv
if [ "${1:-}" != "quick" ]; then
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- python $R/bench.py --workload decode --steps 10 --warmup 3 $B > $O/kt1.log 2>&1
  cp $(find $O/kt1 -name "*kernel_stats.csv" | head -1) $O/rocprof_kernel_stats_configs1.csv
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/bench.py --steps 3 --warmup 1 $B > $O/fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/bench.py --steps 3 --warmup 1 $B > $O/write.log 2>&1
  python $R/tools/pmc_to_json.py $(find $O/fetch -name "*counter_collection.csv" | head -1) \
         $(find $O/write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json full 64 f16x3 "bench.py --steps 3 --warmup 1 $B"
fi
cd $R
rm -rf $O/kt $O/kt1 $O/fetch $O/write
head -8 $O/rocprof_kernel_stats.csv | cut -c1-200
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("default (configs[2]): %.1f img/s %.3f ms/step p50 %s  roofline %s %.1f TFLOP/s frac %.4f" % (
    d["value"], d["ms_per_step"], d["p50_frame_ms_batch1"], d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"]))
for k, v in (d.get("legs") or {}).items():
    print("  leg %-10s %s" % (k, {a: v.get(a) for a in ("value", "ms_per_step", "p50_frame_ms_batch1", "host_fraction", "error") if a in v}))
print("  cpu", d.get("cpu_baseline", {}) and {a: d["cpu_baseline"].get(a) for a in ("value", "cores", "kind")})
PY
ls -la $O
