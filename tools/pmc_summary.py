"""Average the rocprofv3 --pmc counter_collection CSVs under a directory per (kernel substring, counter).
   python tools/pmc_summary.py gpurun_out/prof_pmc_conv igemm16"""
import csv
import glob
import sys
from collections import defaultdict

root, key = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("<")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
by_k = defaultdict(dict)
for (k, c), v in acc.items():
    by_k[k][c] = sum(v) / len(v)
for k, d in by_k.items():
    print(k)
    for c in sorted(d):
        print("  %-34s %16.0f" % (c, d[c]))
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in d:
                print("  %-34s %15.1f%% of wave cycles" % (c, 100 * d[c] / wc))
    bc = d.get("SQ_BUSY_CYCLES")
    if bc:
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INST_CYCLES_VMEM"):
            if c in d:
                print("  %-34s %15.1f%% of SQ busy cycles" % (c, 100 * d[c] / bc))
