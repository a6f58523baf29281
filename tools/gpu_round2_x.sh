#!/bin/bash
# round 2, pass x: concurrent prediction heads at small batch (A/B: cp_set_debug 1048576 = serial heads)
cd /root/repo; mkdir -p gpurun_out/x
timeout 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -3
for b in 1 2 4 8; do for d in 0 1048576; do
timeout 300 python bench.py --batch $b --steps 50 --warmup 5 --dbg $d --no-configs2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('dlav1_34 B=$b dbg $d: %.1f img/s %.3f ms/step p50(B=1 graph) %s'%(d['value'],d['ms_per_step'],d['p50_frame_ms_batch1']))"
done; done | tee gpurun_out/x/heads_ab.txt
for d in 0 1048576; do
timeout 300 python bench.py --workload full --batch 1 --steps 50 --warmup 5 --dbg $d --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('dla_34 full B=1 dbg $d: %.1f img/s %.3f ms/step p50 %s'%(d['value'],d['ms_per_step'],d['p50_frame_ms_batch1']))"
done | tee -a gpurun_out/x/heads_ab.txt
