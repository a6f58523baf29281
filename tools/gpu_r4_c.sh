#!/bin/bash
# round 4: 32-wide halo tile with K halves (cp_set_debug 2 = four waves x 32 pixels), head kernel without spills
set -u
mkdir -p gpurun_out/r04c; O=gpurun_out/r04c/halo_n32_ab.txt; rm -f $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
for d in 2 0 2 0; do echo "dbg $d B=64: $(timeout 300 python bench.py --steps 20 --warmup 3 --no-legs --no-cpu-baseline --no-latency --dbg $d 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(d["value"], "img/s", d["ms_per_step"], "ms/step; offset convs", r["dcn"]["offset_conv_ms"], "ms, head", r["achieved"], "TFLOP/s")')" | tee -a $O; done
for d in 2 0; do echo "dla_34 B=1 $(timeout 200 python tools/lat_probe.py --arch dla_34 --n 300 --dbg $d 2>&1 | tail -1)" | tee -a $O; done
