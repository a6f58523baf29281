#!/bin/bash
# round 2, pass o: N=32 halo kernel with weight fragments straight from L2 (dbg 16384 = LDS-staged tile): parity + step A/B
cd /root/repo; mkdir -p gpurun_out/o
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "halo or backbone or dcn or range_safe" 2>&1 | tail -3
for d in 0 16384 0 16384; do
  timeout 300 python bench.py --dbg $d --no-configs2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('dbg $d: %.1f img/s %.3f ms/step'%(d['value'],d['ms_per_step']), {k:(v['tflops'],v['ms_per_step'],v['launches_per_step']) for k,v in r['all_conv_kernels'].items() if 'halo' in k or 'n32' in k})"
done | tee gpurun_out/o/step_ab.txt
