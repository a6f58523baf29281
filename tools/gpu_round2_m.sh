#!/bin/bash
# round 2, pass m: step A/B of the patch-resident DCN kernel (dbg 32768 = off), both bench configurations
cd /root/repo; mkdir -p gpurun_out/m
for d in 0 32768 0 32768; do
  timeout 300 python bench.py --dbg $d --no-configs2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('dbg $d: %.1f img/s %.3f ms/step'%(d['value'],d['ms_per_step']), {k:(v['tflops'],v['ms_per_step'],v['launches_per_step']) for k,v in r['all_conv_kernels'].items() if 'dcn' in k})"
done | tee gpurun_out/m/step_ab.txt
for d in 0 32768; do
  timeout 300 python bench.py --workload full --dbg $d --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('full dbg $d: %.1f img/s %.3f ms/step'%(d['value'],d['ms_per_step']), {k:(v['tflops'],v['ms_per_step'],v['launches_per_step']) for k,v in r['all_conv_kernels'].items() if 'dcn' in k})"
done | tee -a gpurun_out/m/step_ab.txt
