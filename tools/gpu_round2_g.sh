#!/bin/bash
set -u
O=$PWD/gpurun_out/r2g
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests -m gpu -q -x -k "dcn or backbone_vs_reference or range_safe or spot_parity or stems" 2>&1 | tail -3
python tools/dcn_bench.py
python tools/dcn_bench.py --dbg 2048
python tools/dcn_bench.py --c 128 --co 128 --hw 64
python tools/dcn_bench.py --c 128 --co 128 --hw 64 --dbg 2048
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/tools/dcn_bench.py > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); head -3 $f | cut -c1-150; rm -rf $O/kt
i=0
for c in "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/pmc$i -- python $R/tools/dcn_bench.py --n 3 > $O/pmc$i.log 2>&1
done
cd $R
python tools/pmc_summary.py $O dcn16 > $O/dcn_pmc_summary2.txt 2>&1; cat $O/dcn_pmc_summary2.txt
rm -rf $O/pmc[0-9]
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/bench.json 2>$O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("%.1f img/s  %.3f ms/step  p50 B=1 %.3f ms  roles %s" % (d["value"], d["ms_per_step"], d["p50_frame_ms_batch1"], r["ms_per_step_by_role"]))
c=d["configs2"]; print("configs2 %.1f img/s %.3f ms/step roles %s" % (c["value"], c["ms_per_step"], c["ms_per_step_by_role"]))
PY
