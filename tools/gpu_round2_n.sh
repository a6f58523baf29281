#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/n
python tools/offset_stats.py dlav1_34 2>&1 | tail -20 | tee gpurun_out/n/offset_stats.txt
for d in 0 32768; do
  CP_PROFILE_DUMP=gpurun_out/n/layers_$d.csv timeout 300 python bench.py --dbg $d --steps 8 --warmup 2 --no-configs2 --no-cpu-baseline --no-latency >/dev/null 2>&1
done
python - <<'PY'
import csv
a=[r for r in csv.reader(open('gpurun_out/n/layers_0.csv')) if r[8]=='dcn']
b=[r for r in csv.reader(open('gpurun_out/n/layers_32768.csv')) if r[8]=='dcn']
for x,y in zip(a[:16],b[:16]):
    print('M=%8s N=%4s K=%5s  new %-24s %.4f ms  old %-28s %.4f ms  x%.2f'%(x[1],x[2],x[3],x[0],float(x[6]),y[0],float(y[6]),float(y[6])/float(x[6])))
PY
