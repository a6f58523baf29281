#!/bin/bash
# round 2, pass v: kernel-trace stats of the configs[2] workload (dla_34 B=64 + PnP) at HEAD
cd /root/repo; mkdir -p gpurun_out/v; export TMPDIR=/tmp
R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v/kt -- python $R/bench.py --workload full --steps 10 --warmup 3 --no-cpu-baseline --no-latency > $R/gpurun_out/v/kt.log 2>&1
cd $R
cp $(find gpurun_out/v/kt -name "*kernel_stats.csv" | head -1) gpurun_out/v/kernel_stats_full.csv
rm -rf gpurun_out/v/kt
tail -1 gpurun_out/v/kt.log | cut -c1-300
