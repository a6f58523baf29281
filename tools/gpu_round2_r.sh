#!/bin/bash
# round 2, pass r: halo-resident kernel with weight fragments from L2 on the wide N tiles (experiment): step A/B
cd /root/repo; mkdir -p gpurun_out/r
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "halo" 2>&1 | tail -2
for d in 0 8192 270336 0 270336; do
  timeout 300 python bench.py --dbg $d --no-configs2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('dbg $d: %.1f img/s %.3f ms/step'%(d['value'],d['ms_per_step']), r.get('ms_per_step_by_role'), {k:(v['tflops'],v['ms_per_step'],v['launches_per_step']) for k,v in r['all_conv_kernels'].items() if 'halo' in k or k.startswith('igemm16_f16x3')})"
done | tee gpurun_out/r/step_ab.txt
