"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (every kernel of the run, template arguments kept).
   python tools/pmc_kernels.py <dir> [min_calls]   ->  one block per kernel, derived ratios at the end of each block"""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
# Shader clock of a kernel = per-XCD GRBM_GUI_ACTIVE cycles / the dispatch's duration IN THE SAME PASS (Start / End timestamps of
# the very counter_collection rows that carry GRBM_GUI_ACTIVE).  Round 5 divided the PMC pass's cycles by the durations of a
# separate --kernel-trace run: counter collection lengthens the dispatches, so that quotient read 2.5 - 2.96 GHz on a 2.4 GHz
# part.  (A second argument -- a kernel-trace stats CSV -- is still accepted and only printed as the un-instrumented duration.)
trace_dur = {}
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        k = re.sub(r"\(.*$", "", k).replace("void ", "").strip()
        trace_dur[k] = float(r["AverageNs"])
acc = defaultdict(list)
clk = defaultdict(list)   # kernel -> [(per-XCD GUI cycles, ns)] per dispatch, both from one row
import os

for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    # the same pass's kernel trace (rocprofv3 --kernel-trace --pmc ...: one run, one set of dispatches), joined by dispatch id
    span = {}
    for t in glob.glob(os.path.dirname(f) + "/*kernel_trace.csv"):
        for r in csv.DictReader(open(t)):
            span[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"\(.*$", "", k).replace("void ", "").strip()
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            ns = 0.0
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            elif r.get("Dispatch_Id") in span:
                ns = span[r["Dispatch_Id"]]
            if ns > 0:
                clk[k].append((float(r["Counter_Value"]) / 8, ns))
by_k = defaultdict(dict)
calls = {}
for (k, c), v in acc.items():
    by_k[k][c] = sum(v) / len(v)
    calls[k] = max(calls.get(k, 0), len(v))
keep = ("halo16", "strm16", "pw16", "dcn16", "igemm16", "lowc", "gn_final", "upsample", "peaks", "assoc", "gru_gate", "maxpool")
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
    if clk.get(k):
        cyc, ns = sum(c for c, _ in clk[k]), sum(n for _, n in clk[k])
        mhz = cyc / ns * 1e3
        # GRBM_GUI_ACTIVE of a dispatch also counts a fixed stretch around the kernel (dispatch set-up, counter read-out, cache
        # write-back): for a kernel that is launched at several sizes the SLOPE of cycles over duration is the clock and the
        # intercept that stretch; the plain quotient of a short kernel reads high by intercept / duration
        durs = sorted(n for _, n in clk[k])
        fit = ""
        if len(durs) >= 6 and durs[-1] > 1.5 * durs[0]:
            n_ = len(clk[k])
            mx, my = ns / n_, cyc / n_
            sxx = sum((n - mx) ** 2 for _, n in clk[k])
            sxy = sum((n - mx) * (c - my) for c, n in clk[k])
            slope = sxy / sxx
            fit = "; fit over %d launches of %.0f .. %.0f us: %.0f MHz + %.0f k cycles per launch" % (
                n_, durs[0] / 1e3, durs[-1] / 1e3, slope * 1e3, (my - slope * mx) / 1e3)
        note = "" if mhz <= 2450 else "  (!) above the part's 2.4 GHz: see the fit / the fixed stretch per launch"
        print("  %-34s %15.0f MHz (per-XCD GUI clocks / the same dispatches' own %.1f us in this counter pass%s%s)" % (
            "shader clock", mhz, ns / len(clk[k]) / 1e3, fit, note))
        if k in trace_dur:
            print("  %-34s %15.1f us (kernel-trace run, no counters; the counter pass stretches a dispatch by x%.2f)" % (
                "un-instrumented launch", trace_dur[k] / 1e3, ns / len(clk[k]) / trace_dur[k]))
    elif xcd:
        print("  %-34s %15s (no timestamps for these dispatches in this counter pass: run it with --kernel-trace)" % ("shader clock", "n/a"))
    print()
