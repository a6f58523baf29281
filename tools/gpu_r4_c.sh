#!/bin/bash
# round 4: 64-wide halo tile at three workgroups per CU -- kernel averages from rocprofv3
set -u
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r04c; mkdir -p $O
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency --no-legs > $O/kt3.log 2>&1
python - "$(find $O/kt3 -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("%-90s x%4d avg %8.1f us" % (r['Name'].replace('void (anonymous namespace)::','')[:90], int(r['Calls']), float(r['AverageNs'])/1e3))
PY
tail -1 $O/kt3.log | cut -c1-200
rm -rf $O/kt3
