#!/bin/bash
# round 2, pass q: float32 VALU kernel for the GroupNorm'd heads' final 1x1 (dbg 131072 = matrix-core path): parity + step A/B
cd /root/repo; mkdir -p gpurun_out/q
timeout 900 python -m pytest tests -q -x -m gpu -k "backbone or spot or smoke or detector" 2>&1 | tail -3
for d in 0 131072 0 131072; do
  timeout 300 python bench.py --dbg $d --no-configs2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('dbg $d: %.1f img/s %.3f ms/step'%(d['value'],d['ms_per_step']), r.get('ms_per_step_by_role'))"
done | tee gpurun_out/q/step_ab.txt
