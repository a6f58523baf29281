"""Shader-clock timeline of wave 0 of one mid-launch dcn16s workgroup during its fourth item (tuning build CP_DCN_EXP & 8).
   CENTERPOSE_HIP_LIB=.../libcenterpose_hip_s8.so python tools/dcn16s_timeline.py [--b 64] [--std 1.5]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from centerpose_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--b", type=int, default=64)
ap.add_argument("--std", type=float, default=1.5)
a = ap.parse_args()
hip.set_default_precision("f16x3")
g = torch.Generator().manual_seed(1)
x = torch.randn(a.b, 64, 128, 128, generator=g).cuda()
w = (torch.randn(64, 64, 3, 3, generator=g) / 24).cuda()
bias = torch.randn(64, generator=g).cuda()
off = (torch.randn(a.b, 18, 128, 128, generator=g) * a.std).cuda()
mask = torch.rand(a.b, 9, 128, 128, generator=g).cuda()
L = hip.lib()
L.cp_set_debug(65536 | 2097152)
buf = (ctypes.c_ulonglong * 64)()
runs = []
for it in range(4):
    hip.dcn_v2_forward(x, w, bias, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    torch.cuda.synchronize()
    L.cp_debug_read_dcn16s_clk(buf)
    runs.append(list(buf))
t = runs[-1]
names = {0: "item start", 1: "set-up done", 2: "barrier (exception list)", 3: "chunk-0 corners fetched + blended", 4: "barrier",
         40: "K loop done", 41: "record requested + epilogue issued"}
for c in range(8):
    names[5 + 4 * c] = "chunk %d: next chunk's DMA issued" % c
    names[6 + 4 * c] = "chunk %d: 9 K steps done" % c
    names[7 + 4 * c] = "chunk %d: DMA landed, corners blended" % c
    names[8 + 4 * c] = "chunk %d: barrier" % c
prev = t[0]
for i in sorted(range(64), key=lambda i: t[i]):
    if t[i] == 0:
        continue
    print("%2d  %-40s t = %7d  (+%6d shader clocks)" % (i, names.get(i, "?"), t[i] - t[0], t[i] - prev))
    prev = t[i]
print("item totals of the 4 runs:", [r[41] - r[0] for r in runs])
