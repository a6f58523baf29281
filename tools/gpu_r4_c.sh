#!/bin/bash
# round 4: fused heads -- a workgroup walks every head of its patch (cp_set_debug 2 = one head per workgroup, 1 = slabs)
set -u
mkdir -p gpurun_out/r04c; O=gpurun_out/r04c/head_walk_ab.txt; rm -f $O
for d in 1 2 0 1 2 0; do echo "dbg $d B=64: $(timeout 300 python bench.py --steps 20 --warmup 3 --no-legs --no-cpu-baseline --no-latency --dbg $d 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["value"], "img/s", d["ms_per_step"], "ms/step; head", r["achieved"], "TFLOP/s")')" | tee -a $O; done
for d in 2 0; do echo "dla_34 B=1 $(timeout 200 python tools/lat_probe.py --arch dla_34 --n 300 --dbg $d 2>&1 | tail -1)" | tee -a $O; done
