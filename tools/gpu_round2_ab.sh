#!/bin/bash
# round 2, pass ab: 8-plane previous-heat-map stem through lowc.hip (two groups of 4 planes)
cd /root/repo; mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_tracking_loop.py -q -x -m gpu -k "backbone_vs_reference or stems or track" 2>&1 | tail -3
for w in track track_gru; do
timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('$w: %.1f img/s %.3f ms/step p50 %s'%(d['value'],d['ms_per_step'],d['p50_frame_ms_batch1']), r['ms_per_step_by_role'], {k:(v['tflops'],v['ms_per_step'],v['launches_per_step']) for k,v in r['all_conv_kernels'].items() if 'lowc' in k or 'unaligned' in k})"
done | tee gpurun_out/ab/track.txt
