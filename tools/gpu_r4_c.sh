#!/bin/bash
# round 4: fused heads finished in the kernel (cp_set_debug 1 = slabs + reduction launch)
set -u
mkdir -p gpurun_out/r04c; O=gpurun_out/r04c/head_final_ab.txt; rm -f $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "grouped_fused or backbone_vs_reference or backbone_512" 2>&1 | tail -3
for d in 1 0 1 0; do echo "dbg $d B=64: $(timeout 300 python bench.py --steps 20 --warmup 3 --no-legs --no-cpu-baseline --no-latency --dbg $d 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], "img/s", d["ms_per_step"], "ms/step", d["roofline"]["achieved"], d["roofline"].get("ms_per_launch"))')" | tee -a $O; done
for d in 1 0; do echo "dla_34 B=1 $(timeout 200 python tools/lat_probe.py --arch dla_34 --n 300 --dbg $d 2>&1 | tail -1)" | tee -a $O; done
