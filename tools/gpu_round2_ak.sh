#!/bin/bash
# round 2, pass ak: pnp_kernel with 16 lanes per detection: parity + timing
cd /root/repo; mkdir -p gpurun_out/ak
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_detector.py tests/test_tracking_loop.py -q -x -m gpu -k "pnp or pose or known_answer or schema or track" 2>&1 | tail -3
python tools/pnp_bench.py 2>/dev/null | tee gpurun_out/ak/pnp_bench.txt
timeout 300 python bench.py --workload full --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('full: %.1f img/s %.3f ms/step p50 %s'%(d['value'],d['ms_per_step'],d['p50_frame_ms_batch1']))" | tee -a gpurun_out/ak/pnp_bench.txt
timeout 300 python bench.py --workload full --serial-pnp --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('full --serial-pnp: %.1f img/s %.3f ms/step'%(d['value'],d['ms_per_step']))" | tee -a gpurun_out/ak/pnp_bench.txt
