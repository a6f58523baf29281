"""Shader-clock timeline of one wave of one mid-launch dcn16p block (tuning builds CP_DCN_EXP & 8 [+ 16: per K step]).
   CENTERPOSE_HIP_LIB=.../libcenterpose_hip_dcn8.so python tools/dcn_timeline.py [--b 64] [--std 1.5]"""
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
buf = (ctypes.c_ulonglong * 64)()
runs = []
for it in range(4):
    hip.dcn_v2_forward(x, w, bias, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    torch.cuda.synchronize()
    L.cp_debug_read_dcn_clk(buf)
    runs.append(list(buf))
t = runs[-1]
names = {0: "start", 1: "activation scale read", 2: "zeroing + barrier 1", 3: "records + set-up + barrier 2", 4: "chunk 0 start",
         5: "chunk 0 barrier", 6: "chunk 0 staged", 28: "chunk 1 start", 29: "chunk 1 barrier", 30: "chunk 1 staged",
         60: "K loop done", 61: "epilogue issued", 50: "prologue loads issued", 51: "activation scale arrived",
         52: "first chunk arrived + parked"}
prev = t[0]
for i in sorted(range(64), key=lambda i: t[i]):
    if t[i] == 0:
        continue
    nm = names.get(i) or ("chunk %d step %d done" % ((i - 7) // 24, (i - 7) % 24))
    print("%2d  %-32s t = %7d  (+%6d shader clocks)" % (i, nm, t[i] - t[0], t[i] - prev))
    prev = t[i]
print("totals of the 4 runs:", [r[61] - r[0] for r in runs])
