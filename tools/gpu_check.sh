#!/bin/bash
# One GPU-box pass: the GPU test suite, then (optionally) the artefact refresh.   usage: tools/gpu_check.sh [refresh|quick|none] [pytest -k expr]
set -u
mkdir -p gpurun_out
K="${2:-}"
if [ -n "$K" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x -k "$K" 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.txt
else
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.txt
fi
case "${1:-none}" in
  refresh) bash tools/refresh_profiles.sh ;;
  quick) bash tools/refresh_profiles.sh quick ;;
esac
