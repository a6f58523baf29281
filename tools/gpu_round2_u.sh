#!/bin/bash
# round 2, pass u: where does the batch-1 frame go?  kernel-trace stats of batch-1 steps + per-launch event CSV
cd /root/repo; mkdir -p gpurun_out/u; export TMPDIR=/tmp
CP_PROFILE_DUMP=$PWD/gpurun_out/u/layers_b1.csv python bench.py --batch 1 --steps 64 --warmup 4 --no-cpu-baseline --no-configs2 2>/dev/null | tail -1 > gpurun_out/u/bench_b1.json
R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/u/kt -- python $R/bench.py --batch 1 --steps 200 --warmup 5 --no-cpu-baseline --no-configs2 > $R/gpurun_out/u/kt.log 2>&1
cd $R
cp $(find gpurun_out/u/kt -name "*kernel_stats.csv" | head -1) gpurun_out/u/kernel_stats_b1.csv
rm -rf gpurun_out/u/kt
python -c "
import json;d=json.loads(open('gpurun_out/u/bench_b1.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['p50_frame_ms_batch1'],d['roofline']['ms_per_step_by_role'])"
