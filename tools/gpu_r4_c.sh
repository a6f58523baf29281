#!/bin/bash
# round 4: small launches on 64 x 64 tiles -- final rule, batch 1..8 against the 128-row tiles (cp_set_debug 16)
set -u
mkdir -p gpurun_out/r04c; O=gpurun_out/r04c/t64_ab.txt; rm -f $O
for d in 16 0; do for b in 1 2 4 8; do echo "dbg $d B=$b: $(timeout 200 python bench.py --batch $b --steps 30 --warmup 5 --no-legs --no-cpu-baseline --no-latency --dbg $d 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], "img/s", d["ms_per_step"], "ms/step")')" | tee -a $O; done; done
for a in dla_34 dlav1_34 hourglass; do for d in 16 0; do echo "$a $(timeout 200 python tools/lat_probe.py --arch $a --n 300 --dbg $d 2>&1 | tail -1)" | tee -a $O; done; done
