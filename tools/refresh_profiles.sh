#!/bin/bash
# Regenerate the judged artefacts under gpurun_out/refresh/ on the GPU box (copy into profiles/ afterwards):
#   the default bench line, per-launch layer CSV, rocprofv3 kernel-trace stats and the two PMC passes
#   (FETCH_SIZE / WRITE_SIZE in separate runs, no other tracing) of the default bench command.
set -u
R=$PWD
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
CP_PROFILE_DUMP=$O/layers_default.csv python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-configs2 2>/dev/null | tail -1 > $O/bench_layers.json
python bench.py --precision f32 --no-cpu-baseline --no-configs2 2>/dev/null | tail -1 > $O/bench_f32.json
python bench.py --workload full --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_full.json
python bench.py --workload track --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_track.json
python bench.py --workload track_gru --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_track_gru.json
python bench.py --workload hourglass --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_hourglass.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-configs2 > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-configs2 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-configs2 > $O/write.log 2>&1
cd $R
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/rocprof_kernel_stats.csv
cp $(find $O/fetch -name "*counter_collection.csv" | head -1) $O/pmc_fetch_size.csv
cp $(find $O/write -name "*counter_collection.csv" | head -1) $O/pmc_write_size.csv
python tools/pmc_to_json.py $O/pmc_fetch_size.csv $O/pmc_write_size.csv $O/pmc_traffic.json
rm -rf $O/kt $O/fetch $O/write
head -6 $O/rocprof_kernel_stats.csv | cut -c1-200
for f in default f32 full track track_gru hourglass; do python - <<PY
import json
d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
print("$f: %.1f img/s %.3f ms/step p50 %s" % (d["value"], d["ms_per_step"], d["p50_frame_ms_batch1"]))
PY
done
ls -la $O
