#!/bin/bash
# round 2, pass p: PMC counters of one kernel (substring $KEY) inside the default bench step
set -u
O=$PWD/gpurun_out/p
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
KEY=${KEY:-halo16_kernel}
cd /tmp
i=0
for c in "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU" \
         "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA TA_TA_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/pmc$i -- python $R/bench.py --steps 2 --warmup 1 --no-configs2 --no-cpu-baseline --no-latency > $O/pmc$i.log 2>&1
done
cd $R
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KEY" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
byk=collections.defaultdict(dict)
for (k,c),v in acc.items(): byk[k][c]=(sum(v)/len(v),len(v))
for k,d in byk.items():
    print(k)
    for c in sorted(d): print("  %-30s %14.0f  (n=%d)"%(c,d[c][0],d[c][1]))
PY
rm -rf $O/pmc[0-9]
