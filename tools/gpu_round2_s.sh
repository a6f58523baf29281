#!/bin/bash
# round 2, pass s: quick parity (backbone goldens + spot) and the default bench roles
cd /root/repo; mkdir -p gpurun_out/s
timeout 900 python -m pytest tests -q -x -m gpu -k "${K:-backbone or spot or lowc or stem}" 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --no-configs2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('%.1f img/s %.3f ms/step'%(d['value'],d['ms_per_step']), r.get('ms_per_step_by_role'), {k:(v['tflops'],v['ms_per_step']) for k,v in r['all_conv_kernels'].items() if 'lowc' in k or 'cat' in k})"
done | tee gpurun_out/s/step.txt
