#!/bin/bash
# round 2, pass aa: fused level0 -> level1 kernel: parity + A/B (cp_set_debug 2097152 = two kernels)
cd /root/repo; mkdir -p gpurun_out/aa
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "fused_level0 or backbone or spot" 2>&1 | tail -5
for d in 0 2097152 0 2097152; do
timeout 300 python bench.py --dbg $d --no-configs2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('dbg $d: %.1f img/s %.3f ms/step'%(d['value'],d['ms_per_step']), r['ms_per_step_by_role']['lowc'], {k:(v['tflops'],v['ms_per_step']) for k,v in r['all_conv_kernels'].items() if 'lowc' in k})"
done | tee gpurun_out/aa/lowc_pair_ab.txt
