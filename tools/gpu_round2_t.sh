#!/bin/bash
# round 2, pass t: full GPU test suite + refresh of every judged artefact at HEAD
cd /root/repo; mkdir -p gpurun_out/t
( echo "# pytest -m gpu on MI355X (gpurun call T, round 2)"; timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -4; echo "rc $?" ) | tee gpurun_out/t/pytest_gpu.txt
timeout 1500 bash tools/refresh_profiles.sh 2>&1 | tail -25
