#!/bin/bash
# round 2, pass ac: pw16.hip (1x1 layers as a register-only stream): parity + A/B (cp_set_debug 4194304 = LDS-staged loop)
cd /root/repo; mkdir -p gpurun_out/ac
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "pointwise or range_safe or backbone or spot or both_precisions" 2>&1 | tail -5
for d in 0 4194304 0 4194304; do
timeout 300 python bench.py --dbg $d --no-configs2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('dbg $d: %.1f img/s %.3f ms/step'%(d['value'],d['ms_per_step']), 'conv1x1', r['ms_per_step_by_role']['conv1x1'], {k:(v['tflops'],v['ms_per_step'],v['launches_per_step']) for k,v in r['all_conv_kernels'].items() if 'pw16' in k or k.startswith('igemm16')})"
done | tee gpurun_out/ac/pw16_ab.txt
