#!/bin/bash
# round-4 GPU pass A: LDS-DMA probe, dcn16s parity tests, dcn16p / dcn16s A/B per layer shape
set -u
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -Wno-inline-asm tools/probe/dma_probe.hip -o /tmp/dma_probe && timeout 60 /tmp/dma_probe 2>&1 | tee gpurun_out/dma_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "streamed" 2>&1 | tail -15 | tee gpurun_out/pytest_streamed.txt
cd /tmp && rm -rf /tmp/ab && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab -- python $R/tools/dcn_ab.py --b ${B:-64} > $R/gpurun_out/dcn_ab_run.txt 2>&1
cd $R && tail -8 gpurun_out/dcn_ab_run.txt; python tools/dcn_ab.py --parse /tmp/ab --b ${B:-64} 2>&1 | tee gpurun_out/dcn_ab.txt
