#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/ag
CP_PROFILE_DUMP=$PWD/gpurun_out/ag/layers_hg.csv python bench.py --workload hourglass --steps 4 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | cut -c1-120
