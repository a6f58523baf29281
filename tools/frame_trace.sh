#!/bin/bash
# Kernel trace of the batch-1 frame (cp_model_detect replayed from its hipGraph): launches per frame and where the frame's
# microseconds go.  usage: tools/frame_trace.sh [arch ...]  -> gpurun_out/frame_trace_<arch>.txt
set -u
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out
for arch in "${@:-dla_34 dlav1_34}"; do
 for a in $arch; do
  O=$R/gpurun_out/ft_$a; rm -rf $O
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/tools/lat_probe.py --arch $a --n 200 > $O.log 2>&1 )
  python - "$(find $O -name '*kernel_stats.csv' | head -1)" $a "$(grep p50 $O.log | tail -1)" <<'PY' | tee $R/gpurun_out/frame_trace_$a.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frames = 205.0
keep = [r for r in rows if float(r["Calls"]) >= frames * 0.9]
tot = sum(float(r["TotalDurationNs"]) for r in keep) / frames / 1e3
n = sum(float(r["Calls"]) for r in keep) / frames
print("%s: %s" % (sys.argv[2], sys.argv[3]))
print("launches per frame %.1f, kernel time per frame %.1f us (sum of kernel durations; the frame's wall time also holds the gaps)" % (n, tot))
for r in sorted(keep, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:78]
    print("  %-80s x%5.1f  avg %7.1f us  %6.1f us/frame" % (name, float(r["Calls"]) / frames, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / frames / 1e3))
PY
  rm -rf $O
 done
done
