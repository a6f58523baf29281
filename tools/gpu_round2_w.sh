#!/bin/bash
# round 2, pass w: PoseStage test; configs[2] with the PnP on the side stream vs serial; batch-1 frame after the split-K epilogue change
cd /root/repo; mkdir -p gpurun_out/w
timeout 600 python -m pytest tests/test_gpu_detector.py -q -x -m gpu -k "pose_stage or known_answer or schema" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "backbone or spot" 2>&1 | tail -2
for f in "" "--serial-pnp" "" "--serial-pnp"; do
timeout 300 python bench.py --workload full --no-cpu-baseline $f 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('full $f: %.1f img/s %.3f ms/step p50 %s'%(d['value'],d['ms_per_step'],d['p50_frame_ms_batch1']))"
done | tee gpurun_out/w/full_ab.txt
timeout 300 python bench.py --no-configs2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('default: %.1f img/s %.3f ms/step p50 %s'%(d['value'],d['ms_per_step'],d['p50_frame_ms_batch1']))" | tee gpurun_out/w/default.txt
