#!/bin/bash
# round 2, pass j: patch-resident DCN kernel (dcn16p.hip): parity, micro-benchmark, step A/B
cd /root/repo; mkdir -p gpurun_out/j
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "dcn" 2>&1 | tail -15 > gpurun_out/j/pytest_dcn.txt
cat gpurun_out/j/pytest_dcn.txt
for d in 0 32768; do
  timeout 120 python tools/dcn_bench.py --dbg $d --std 1.5 2>&1 | tail -1
  timeout 120 python tools/dcn_bench.py --dbg $d --std 1.5 --c 128 --co 128 --hw 64 2>&1 | tail -1
  timeout 120 python tools/dcn_bench.py --dbg $d --std 1.5 --c 256 --co 256 --hw 32 2>&1 | tail -1
  timeout 120 python tools/dcn_bench.py --dbg $d --std 4.0 2>&1 | tail -1
done | tee gpurun_out/j/dcn_bench.txt
for d in 0 32768 0 32768; do
  timeout 300 python bench.py --dbg $d --no-configs2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('dbg $d: %.1f img/s %.3f ms/step'%(d['value'],d['ms_per_step']), {k:(v['tflops'],v['ms_per_step']) for k,v in r['all_conv_kernels'].items() if 'dcn' in k})"
done | tee gpurun_out/j/step_ab.txt
