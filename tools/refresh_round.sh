#!/bin/bash
# A round's artefacts on the GPU box -> gpurun_out/<tag>/ (copied into profiles/<tag>_* afterwards):
#   full GPU test suite, the default bench line (+ detail file), rocprofv3 kernel-trace stats of the bench command, the SQ
#   counter passes (tools/pmc_sq.txt, one --pmc line per run) and the FETCH_SIZE / WRITE_SIZE passes, the DCN A/B per layer
#   shape, the batch-1 frame trace, the PnP micro-benchmark.    usage: tools/refresh_round.sh <tag, e.g. r05> [quick]
set -u
TAG=${1:?usage: tools/refresh_round.sh <tag> [quick]}; shift
R=$PWD; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
CP_BENCH_DETAIL=$O/bench_detail.json timeout 1200 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
B="--no-cpu-baseline --no-latency --no-legs"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 10 --warmup 3 $B > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/rocprof_kernel_stats.csv
rm -rf $O/kt
if [ "${1:-}" != "quick" ]; then
  CP_PROFILE_DUMP=$O/layers_default.csv timeout 300 python $R/bench.py --steps 8 --warmup 2 $B 2>/dev/null | tail -1 > $O/bench_layers.json
  i=0
  grep "^pmc:" $R/tools/pmc_sq.txt | while read -r _ ctrs; do
    i=$((i+1))
    timeout 400 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/sq$i -- python $R/bench.py --steps 3 --warmup 1 $B > $O/sq$i.log 2>&1
  done
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -- python $R/bench.py --steps 3 --warmup 1 $B > $O/kt2.log 2>&1
  python $R/tools/pmc_kernels.py $O $(find $O/kt2 -name "*kernel_stats.csv" | head -1) > $O/pmc_sq_counters.txt 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/bench.py --steps 3 --warmup 1 $B > $O/fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/bench.py --steps 3 --warmup 1 $B > $O/write.log 2>&1
  python $R/tools/pmc_to_json.py $(find $O/fetch -name "*counter_collection.csv" | head -1) \
         $(find $O/write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json full 64 f16x3 "bench.py --steps 3 --warmup 1 $B"
  rm -rf $O/sq1 $O/sq2 $O/sq3 $O/kt2 $O/fetch $O/write
  rm -rf /tmp/ab && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab -- python $R/tools/dcn_ab.py --b 64 > $O/dcn_ab_run.txt 2>&1
  (cd $R && python tools/dcn_ab.py --parse /tmp/ab --b 64 > $O/dcn_ab.txt 2>&1)
  (cd $R && timeout 300 python tools/pnp_bench.py > $O/pnp_bench.txt 2>&1)
  (cd $R && hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/probe/pk_opsel_lds_hazard.hip -o /tmp/hz 2>/dev/null && timeout 120 /tmp/hz 20000 > $O/pk_opsel_lds_hazard.txt 2>&1)
  (cd $R && timeout 400 bash tools/frame_trace.sh dla_34 > /dev/null 2>&1; cp gpurun_out/frame_trace_dla_34.txt $O/ 2>/dev/null)
fi
cd $R
head -6 $O/rocprof_kernel_stats.csv | cut -c1-180
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("line: %d bytes; default (configs[2]): %.1f img/s %.3f ms/step p50 %s / %s  roofline %s %.1f TFLOP/s frac %.4f dcn %s" % (
    len(json.dumps(d)), d["value"], d["ms_per_step"], d["p50_frame_ms_batch1"], d.get("p50_frame_ms_batch1_network_decode"),
    d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"].get("dcn")))
for k, v in (d.get("legs") or {}).items():
    print("  leg %-12s %s" % (k, json.dumps(v)[:230]))
print("  cpu", d.get("cpu_baseline"))
PY
ls $O
