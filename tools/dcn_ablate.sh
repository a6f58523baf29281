#!/bin/bash
# Timing ablations of dcn16p_kernel on the heaviest DCN layer (64 -> 64 @ 128 x 128): rocprofv3 average of the kernel for
# each cp_set_debug ablation mask (dcn16p.hip: ABL).  usage: tools/dcn_ablate.sh [B] ; output gpurun_out/dcn_ablate.txt
set -u
R=$PWD; O=$R/gpurun_out/abl; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
B=${1:-64}
BASE=$((1<<30))
cd /tmp
ALL=$((BASE|1<<25|1<<26|1<<27|1<<28|1<<29))
LIST="${MASKS:-0 $BASE $((BASE|1<<25)) $((BASE|1<<26)) $((BASE|1<<27)) $((BASE|1<<28)) $((BASE|1<<29)) $((BASE|1<<25|1<<26)) $((BASE|1<<25|1<<26|1<<27)) $((BASE|1<<25|1<<26|1<<28|1<<29)) $ALL $((ALL|1<<31)) $((BASE|1<<31))}"
for m in $LIST; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r$m -- python $R/tools/dcn_bench.py --b $B --n 6 --std ${STD:-1.5} --dbg $m > $O/log$m.txt 2>&1
  f=$(find $O/r$m -name "*kernel_stats.csv" | head -1)
  python - "$f" "$m" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dcn16p_kernel" in r["Name"]:
        m = (int(sys.argv[2]) & 0xffffffff) >> 25
        tags = [n for i, n in enumerate(("no-gather", "no-blend", "no-mfma", "no-weights", "no-staging", "abl-variant", "no-epilogue")) if m >> i & 1]
        tags += [n for i, n in enumerate(("no-record-loads", "no-setup-math", "no-zeroing")) if int(sys.argv[2]) >> i & 1]
        print("mask %-45s dcn16p avg %8.1f us  (%s calls)" % ("+".join(tags) or "production kernel", float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
  rm -rf $O/r$m
done | tee $R/gpurun_out/dcn_ablate.txt
