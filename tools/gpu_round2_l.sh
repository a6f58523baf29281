#!/bin/bash
# round 2, pass l: dcn16p iteration -- parity of the DCN tests + kernel time on the micro-benchmark shapes
cd /root/repo; mkdir -p gpurun_out/l; export TMPDIR=/tmp
[ -n "${SKIP_TESTS:-}" ] || timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "dcn" 2>&1 | tail -3
R=$PWD; cd /tmp
for args in "--std 1.5 --dbg 65536" "--std 0.5 --dbg 65536" "--std 2.0 --dbg 65536" "--std 1.5 --c 128 --co 128 --hw 64 --dbg 65536" "--std 1.5 --c 256 --co 256 --hw 32 --dbg 65536" "--std 1.5 --dbg 32768" "--std 1.5 --c 128 --co 128 --hw 64 --dbg 32768"; do
  rm -rf $R/gpurun_out/l/kt
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/l/kt -- python $R/tools/dcn_bench.py --n 5 $args > $R/gpurun_out/l/kt.log 2>&1
  f=$(find $R/gpurun_out/l/kt -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'dcn16' in r['Name']: print('$args:', r['Name'][28:60], r['Calls'], '%.1f us' % (float(r['AverageNs'])/1e3))"
done | tee $R/gpurun_out/l/kernel_times.txt
rm -rf $R/gpurun_out/l/kt
