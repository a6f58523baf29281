#!/bin/bash
# the default bench line again (box-to-box spread of the power-limited step)
cd /root/repo; mkdir -p gpurun_out/ae
python bench.py 2>/dev/null | tail -1 > gpurun_out/ae/bench_default.json
python -c "
import json;d=json.loads(open('gpurun_out/ae/bench_default.json').read());r=d['roofline']
print(d['value'],d['ms_per_step'],d['p50_frame_ms_batch1'],r['kernel'],r['achieved'],r['frac'],d['configs2']['value'])"
