#!/bin/bash
# round 2, pass k: PMC counters of the patch-resident DCN kernel on the 64->64 @128x128 B=32 micro-benchmark
set -u
O=$PWD/gpurun_out/k
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/tools/dcn_bench.py --n 5 --std ${STD:-1.5} --dbg ${DBG:-65536} > $O/kt.log 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); head -4 $f | cut -c1-170; rm -rf $O/kt
i=0
for c in "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" \
         "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/pmc$i -- python $R/tools/dcn_bench.py --n 3 --std ${STD:-1.5} --dbg ${DBG:-65536} > $O/pmc$i.log 2>&1
done
cd $R
python tools/pmc_summary.py $O dcn16p > $O/dcn16p_pmc.txt 2>&1; cat $O/dcn16p_pmc.txt
rm -rf $O/pmc[0-9]
