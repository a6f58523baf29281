"""A/B of the DCNv2 main kernels per layer shape of dla_34 at batch B: dcn16p (patch-resident, 64-wide N tile), dcn16pw (the same on
the 128-wide N tile where the layer has whole 128-channel tiles) and dcn16s (persistent, streamed).
Run under rocprofv3 --kernel-trace; `--parse DIR` then prints the average kernel duration per (shape, kernel) from the trace.
usage: rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/dcn_ab.py [--b 64] [--n 5] ; python tools/dcn_ab.py --parse OUT"""
import argparse
import csv
import glob
import os
import sys

SHAPES = [  # (Cin, Cout, HW, count in the network)
    (64, 64, 128, 5), (128, 64, 64, 4), (128, 128, 64, 2), (256, 128, 32, 2), (256, 256, 32, 1), (256, 64, 32, 1), (512, 256, 16, 1)]
MODES = [("dcn16p", 1048576 | 524288 | 67108864), ("dcn16pw", 1048576 | 67108864), ("dcn16s", 2097152), ("dcn16t", 33554432)]  # (dcn16pw: the 128-wide N tile where Cout % 128 == 0)

ap = argparse.ArgumentParser()
ap.add_argument("--b", type=int, default=64)
ap.add_argument("--n", type=int, default=5)
ap.add_argument("--std", type=float, default=1.5)
ap.add_argument("--parse", default=None)
ap.add_argument("--nshapes", type=int, default=7)
ap.add_argument("--only", default=None, help="dcn16p | dcn16pw | dcn16s | dcn16t")
a = ap.parse_args()
SHAPES = SHAPES[:a.nshapes]
if a.only:
    MODES = [m for m in MODES if m[0] == a.only]

if a.parse:
    f = sorted(glob.glob(os.path.join(a.parse, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("dcn16p_kernel", "dcn16s_kernel", "dcn16t_kernel"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    i = 0
    tot = {m: 0.0 for m, _ in MODES}
    for (ci, co, hw, cnt) in SHAPES:
        line = "%3d->%3d @%3d x%d:" % (ci, co, hw, cnt)
        for m, _ in MODES:
            grp = rows[i:i + 2 + a.n]
            i += 2 + a.n
            names = {("dcn16t" if "dcn16t" in r["Kernel_Name"] else "dcn16s" if "dcn16s" in r["Kernel_Name"] else "dcn16pw" if "dcn16p_kernel<4" in r["Kernel_Name"] else "dcn16p") for r in grp}
            d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp[2:]]
            avg = sum(d) / max(1, len(d))
            tot[m] += avg * cnt
            line += "  %s %8.1f us (%s)" % (m, avg, "/".join(sorted(names)))
        print(line)
    print("network DCN main total per step: " + "  ".join("%s %.3f ms" % (m, tot[m] / 1e3) for m, _ in MODES))
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from centerpose_amd import hip  # noqa: E402

hip.set_default_precision("f16x3")
g = torch.Generator().manual_seed(1)
for (ci, co, hw, cnt) in SHAPES:
    x = torch.randn(a.b, ci, hw, hw, generator=g).cuda()
    w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).cuda()
    bias = torch.randn(co, generator=g).cuda()
    off = (torch.randn(a.b, 18, hw, hw, generator=g) * a.std).cuda()
    mask = torch.rand(a.b, 9, hw, hw, generator=g).cuda()
    outs = []
    for m, dbg in MODES:
        hip.lib().cp_set_debug(65536 | dbg)
        for _ in range(2 + a.n):
            y = hip.dcn_v2_forward(x, w, bias, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1)
        torch.cuda.synchronize()
        outs.append(y)
    hip.lib().cp_set_debug(0)
    err = float((outs[0] - outs[2]).abs().max() / outs[0].abs().max()) if len(outs) > 2 else -1.0
    errt = float((outs[0] - outs[-1]).abs().max() / outs[0].abs().max())
    print("%d->%d @%d: max |dcn16p - dcn16s| / max = %.2e; |dcn16p - dcn16t| / max = %.2e; dcn16t == dcn16s bit for bit: %s; 128-wide == 64-wide: %s" % (
        ci, co, hw, err, errt, bool(torch.equal(outs[2], outs[-1])) if len(outs) > 3 else "-", bool(torch.equal(outs[0], outs[1])) if len(outs) > 2 else "-"), flush=True)
