#!/bin/bash
# round 2, pass z: GroupNorm'd head finals on a side stream under the next head's 3x3 (A/B: cp_set_debug 1048576 = one stream)
cd /root/repo; mkdir -p gpurun_out/z
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "backbone or spot or gru" 2>&1 | tail -3
for b in 32 8; do for d in 0 1048576 0 1048576; do
timeout 300 python bench.py --batch $b --dbg $d --no-configs2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('B=$b dbg $d: %.1f img/s %.3f ms/step'%(d['value'],d['ms_per_step']), r['ms_per_step_by_role'])"
done; done | tee gpurun_out/z/side_finals_ab.txt
timeout 300 python bench.py --workload track_gru --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | cut -c1-200
