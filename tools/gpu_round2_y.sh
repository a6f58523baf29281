#!/bin/bash
# round 2, pass y: ConvGRU gate arithmetic on hardware exp2 / rcp
cd /root/repo; mkdir -p gpurun_out/y
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "gru or GRU or backbone" 2>&1 | tail -3
for d in 0 0; do
timeout 300 python bench.py --dbg $d --no-configs2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('dbg $d: %.1f img/s %.3f ms/step'%(d['value'],d['ms_per_step']), r['ms_per_step_by_role']['gru'], {k:(v['tflops'],v['ms_per_step']) for k,v in r['all_conv_kernels'].items() if 'gru' in k})"
done | tee gpurun_out/y/gru_ab.txt
