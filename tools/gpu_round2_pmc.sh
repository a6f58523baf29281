#!/bin/bash
# round 2: SQ counters (issue / wait / MFMA / LDS) of every kernel of the default bench step, three --pmc passes
cd /root/repo; mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
R=$PWD; cd /tmp
i=0
while read -r line; do
  i=$((i+1))
  rocprofv3 --pmc ${line#pmc: } --output-format csv -d $R/gpurun_out/pmc/p$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-configs2 > $R/gpurun_out/pmc/p$i.log 2>&1
  tail -1 $R/gpurun_out/pmc/p$i.log | cut -c1-120
done < $R/tools/pmc_sq.txt
cd $R
python tools/pmc_kernels.py gpurun_out/pmc profiles/r02_rocprof_kernel_stats.csv > gpurun_out/pmc/sq_counters.txt
tar czf gpurun_out/pmc/raw.tgz -C gpurun_out/pmc p1 p2 p3; rm -rf gpurun_out/pmc/p1 gpurun_out/pmc/p2 gpurun_out/pmc/p3
head -50 gpurun_out/pmc/sq_counters.txt
