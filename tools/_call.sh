python -m pytest tests/test_gpu_parity.py -m gpu -q -k "grouped or spot_parity or backbone_vs_reference_golden" 2>&1 | tail -3
B="--no-cpu-baseline --no-legs"
for d in 0 16 8; do python bench.py $B --dbg $d 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg $d', d['value'], d['ms_per_step'], d['p50_frame_ms_batch1'], d['roofline']['achieved'], d['roofline']['ms_per_step_by_role'])"; done
