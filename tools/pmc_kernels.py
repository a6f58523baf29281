"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (every kernel of the run, template arguments kept).
   python tools/pmc_kernels.py <dir> [min_calls]   ->  one block per kernel, derived ratios at the end of each block"""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
# optional: a rocprofv3 --kernel-trace --stats CSV of the same command -> shader clock = per-XCD GUI cycles / average duration
dur = {}
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        k = re.sub(r"\(.*$", "", k).replace("void ", "").strip()
        dur[k] = float(r["AverageNs"])
acc = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"\(.*$", "", k).replace("void ", "").strip()
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
by_k = defaultdict(dict)
calls = {}
for (k, c), v in acc.items():
    by_k[k][c] = sum(v) / len(v)
    calls[k] = max(calls.get(k, 0), len(v))
keep = ("halo16", "dcn16", "igemm16", "lowc", "gn_final", "upsample", "peaks", "assoc", "gru_gate", "maxpool")
for k in sorted(by_k, key=lambda k: -by_k[k].get("GRBM_GUI_ACTIVE", 0) * calls[k]):
    if not any(s in k for s in keep):
        continue
    d = by_k[k]
    print("%s   (%d dispatches sampled)" % (k, calls[k]))
    for c in sorted(d):
        print("  %-34s %16.0f" % (c, d[c]))
    wc, bc, gui = d.get("SQ_WAVE_CYCLES"), d.get("SQ_BUSY_CYCLES"), d.get("GRBM_GUI_ACTIVE")
    if wc:
        for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS"):
            if c in d:
                print("  %-34s %15.1f%% of wave cycles" % (c, 100 * d[c] / wc))
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES / SQ_LDS_IDX_ACTIVE are summed over every SIMD / CU
    # in shader clocks (checked on halo16: BUSY_CYCLES = 32 x SQ_INSTS_MFMA, MOPS x 512 = 3 x the algorithmic FLOPs)
    xcd = gui / 8 if gui else None
    if xcd and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        print("  %-34s %15.1f%% of (kernel clocks x 1024 SIMDs)" % ("MFMA pipe busy", 100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (xcd * 1024)))
    if xcd and "SQ_LDS_IDX_ACTIVE" in d:
        print("  %-34s %15.1f%% of (kernel clocks x 256 CUs)" % ("LDS pipe active", 100 * d["SQ_LDS_IDX_ACTIVE"] / (xcd * 256)))
        if "SQ_LDS_BANK_CONFLICT" in d:
            print("  %-34s %15.1f%% of (kernel clocks x 256 CUs)" % ("LDS bank-conflict cycles", 100 * d["SQ_LDS_BANK_CONFLICT"] / (xcd * 256)))
    if xcd and "SQ_INSTS_VALU_MFMA_MOPS_F16" in d:
        # 1 MOP = 512 f16 FLOPs; the 2.5 PFLOP/s peak at 2.4 GHz is 1.0417e6 FLOP per clock
        print("  %-34s %15.1f%% of the f16 matrix peak per kernel clock" % ("MFMA MOPS", 100 * d["SQ_INSTS_VALU_MFMA_MOPS_F16"] * 512 / (xcd * 1.0417e6)))
    if xcd and k in dur:
        print("  %-34s %15.0f MHz (per-XCD GUI clocks / %.1f us average launch of the kernel-trace run)" % ("shader clock", xcd / dur[k] * 1e3, dur[k] / 1e3))
    print()
