#!/bin/bash
# round-4 GPU pass B: full GPU suite, DCN stress, dcn16p / dcn16s A/B, short bench
set -u
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.txt
timeout 200 python tools/probe/dcn16p_race.py dcn16p 3000 16 | tail -3
timeout 200 python tools/probe/dcn16p_race.py dcn16s 1500 16 | tail -2
cd /tmp && rm -rf /tmp/ab && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab -- python $R/tools/dcn_ab.py --b 64 > $R/gpurun_out/dcn_ab_run.txt 2>&1
cd $R && grep "max |" gpurun_out/dcn_ab_run.txt; python tools/dcn_ab.py --parse /tmp/ab --b 64 2>&1 | tee gpurun_out/dcn_ab.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_quick.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_quick.json').read())
print('bench: %.1f img/s %.3f ms/step' % (d['value'], d['ms_per_step']), {k: d['roofline'].get(k) for k in ('kernel','achieved','frac')}, d['roofline'].get('dcn'))
PY
