#!/bin/bash
# round 4: IDAUp proj DCNs on a second stream at batch 1-2 (cp_set_debug 2 = one stream)
set -u
mkdir -p gpurun_out/r04c; O=gpurun_out/r04c/two_stream_ab.txt; rm -f $O
for a in dla_34; do for d in 2 0; do echo "$a $(timeout 200 python tools/lat_probe.py --arch $a --n 300 --dbg $d 2>&1 | tail -1)" | tee -a $O; echo "$a eager $(timeout 200 python tools/lat_probe.py --arch $a --n 300 --dbg $d --eager 2>&1 | tail -1)" | tee -a $O; done; done
for d in 2 0; do for b in 1 2; do echo "dbg $d B=$b: $(timeout 200 python bench.py --batch $b --steps 30 --warmup 5 --no-legs --no-cpu-baseline --no-latency --dbg $d 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], "img/s", d["ms_per_step"], "ms/step")')" | tee -a $O; done; done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_detector.py -q -x 2>&1 | tail -3
