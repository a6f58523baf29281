#!/bin/bash
# round-2 GPU call B: DCN kernel A/B (new pipelined gather vs previous loop vs alternative wave counts), parity of the
# new kernel, batch-1 latency A/B of the |max| tracking.
set -u
O=gpurun_out/r2b
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "dcn or backbone_vs_reference or range_safe or small_weights or fused_head or detect_one_call" > $O/pytest.txt 2>&1; echo "rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
for d in 0 1024 2048 512; do
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-configs2 --dbg $d > $O/bench_dbg$d.json 2>$O/bench_dbg$d.err
  python - <<PY
import json
d=json.loads(open("$O/bench_dbg$d.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("dbg $d: %.1f img/s  %.3f ms/step  p50 B=1 %.3f ms  dcn %s  roles %s" % (d["value"], d["ms_per_step"], d["p50_frame_ms_batch1"], {k:r["dcn"][k] for k in ("main_ms","offset_conv_ms","hbm_gbps")}, r["ms_per_step_by_role"]))
print({k:(v["tflops"],v["ms_per_step"]) for k,v in r["all_conv_kernels"].items() if "dcn" in k})
PY
done
