#!/bin/bash
# Timing ablations of dcn16s_kernel (64 -> 64 @ 128 x 128, B = 64): rocprofv3 kernel-trace average per variant library
# (built with: make -C centerpose_amd/csrc variant VAR=a<bits> FILES=dcn16s DEFS=-DCP_DCN_EXP=<bits>)
set -u
R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out; : > gpurun_out/dcn16s_ablate.txt
name() { case $1 in 0) echo "production";; 1048576) echo "no DMA";; 2097152) echo "no weight loads";; 3145728) echo "no DMA, no weight loads";; 4194304) echo "no epilogue stores";; 8388608) echo "no gather reads";; 16777216) echo "no MFMAs";; 33554432) echo "no blend";; 62914560) echo "no gather, MFMA, blend, weights";; *) echo "bits $1";; esac; }
for v in 0 1048576 2097152 3145728 4194304 8388608 16777216 33554432 62914560; do
  lib=$R/centerpose_amd/libcenterpose_hip_a$v.so; [ $v = 0 ] && lib=$R/centerpose_amd/libcenterpose_hip.so
  [ -f $lib ] || continue
  cd /tmp && rm -rf /tmp/abl && CENTERPOSE_HIP_LIB=$lib timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/abl -- python $R/tools/dcn_ab.py --b 64 --nshapes 1 --only dcn16s > /tmp/abl.log 2>&1
  cd $R && echo "$(name $v): $(python tools/dcn_ab.py --parse /tmp/abl --b 64 --nshapes 1 --only dcn16s | head -1)" | tee -a gpurun_out/dcn16s_ablate.txt
done
