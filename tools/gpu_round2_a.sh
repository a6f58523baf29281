#!/bin/bash
# round-2 GPU call A: parity tests (incl. the new range-safety cases), the diagnostic sweep, the default bench line
# and the clock / data-dependent-power probes behind DESIGN 3.1's "power-limited" claim.
set -u
O=gpurun_out/r2a
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -x --deselect tests/test_gpu_parity.py::test_backbone_at_bench_batch_spot_parity > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
timeout 600 python -m pytest tests -m gpu -q -k "bench_batch_spot or range_safe or small_weights or wide_dynamic" > $O/pytest_range.txt 2>&1
echo "pytest-range rc $?" >> $O/pytest_range.txt
tail -5 $O/pytest_range.txt
timeout 600 python tests/gpu_diag.py > $O/diag_f16x3.txt 2>&1; tail -3 $O/diag_f16x3.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
CENTERPOSE_HIP_LIB=$PWD/centerpose_amd/libcenterpose_hip_exp256.so timeout 120 python tools/probe/clk_test.py > $O/clock_probe.txt 2>&1
timeout 120 python tools/probe/data_power_test.py >> $O/clock_probe.txt 2>&1
cat $O/clock_probe.txt
