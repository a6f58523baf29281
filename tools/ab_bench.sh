#!/bin/bash
# A/B of library builds on one GPU box: alternating short bench runs, per-kernel ms per step side by side.
#   usage: tools/ab_bench.sh OUTDIR REPS VARIANT...   ("" = the product library, "_x" = centerpose_amd/libcenterpose_hip_x.so,
#                                                      "@N" = the product library under cp_set_debug N: bench.py --dbg N)
set -u
R=$PWD; O=$1; REPS=$2; shift 2; mkdir -p $O; export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    n=${v:-_prod}; lib=$v; dbg=0
    case "$v" in @*) lib=""; dbg=${v#@};; esac
    CP_BENCH_DETAIL=$O/detail$n.$rep.json CENTERPOSE_HIP_LIB=$R/centerpose_amd/libcenterpose_hip$lib.so \
      timeout 300 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-latency --dbg $dbg > $O/line$n.$rep.json 2>/dev/null
  done
done
python - $O $REPS "$@" <<'PY'
import json, sys, glob
O, reps, libs = sys.argv[1], int(sys.argv[2]), [v or "_prod" for v in sys.argv[3:]]
rows = {}
for n in libs:
    for r in range(1, reps + 1):
        try:
            d = json.load(open("%s/detail%s.%d.json" % (O, n, r)))
        except Exception as e:
            print("missing", n, r, e); continue
        k = d["roofline"]["all_conv_kernels"]
        rows.setdefault("TOTAL ms/step", {}).setdefault(n, []).append(d["ms_per_step"])
        rows.setdefault("conv ms/step", {}).setdefault(n, []).append(d["roofline"]["conv_ms_per_step"])
        for name, v in k.items():
            rows.setdefault(name, {}).setdefault(n, []).append(v["ms_per_step"])
print("%-34s" % "kernel (ms per step, runs)" + "".join("%-30s" % n for n in libs))
for name, per in rows.items():
    print("%-34s" % name + "".join("%-30s" % " ".join("%.3f" % x for x in per.get(n, [])) for n in libs))
PY
