"""Shader-clock timeline of wave 0 of one mid-launch halo16 workgroup (tuning build -DCP_HALO_STAMP):
   make -C centerpose_amd/csrc variant VAR=hstamp FILES=halo16 DEFS=-DCP_HALO_STAMP
   CENTERPOSE_HIP_LIB=centerpose_amd/libcenterpose_hip_hstamp.so python tools/halo16_timeline.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from centerpose_amd import hip  # noqa: E402

hip.set_default_precision("f16x3")
L = hip.lib()
buf = (ctypes.c_ulonglong * 32)()
NAMES = {0: "start", 20: "last K loop done", 21: "epilogue done"}
for ch in range(4):
    NAMES[1 + 4 * ch] = "chunk %d: barrier passed (previous K loop over)" % ch
    NAMES[2 + 4 * ch] = "chunk %d: this wave's share staged" % ch
    NAMES[3 + 4 * ch] = "chunk %d: barrier passed, K loop starts" % ch
for cin, cout, hw, what in ((128, 128, 64, "BasicBlock 128 -> 128 @64^2 (N = 128, 2 chunks)"),
                            (256, 256, 32, "BasicBlock 256 -> 256 @32^2 (N = 128, 4 chunks, 2 N tiles)"),
                            (64, 64, 128, "BasicBlock 64 -> 64 @128^2 (N = 64)"),
                            (64, 27, 128, "offset convolution 64 -> 27 @128^2 (N = 32)")):
    x = torch.randn(64, hw, hw, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    runs = []
    for it in range(4):
        hip.conv2d_nhwc(x, w, None, None, None, 1, 1, 1)
        torch.cuda.synchronize()
        assert L.cp_debug_read_halo_clk(buf) == 0
        runs.append([int(v) for v in buf])
    t = runs[-1]
    print(what)
    prev = t[0]
    for i in sorted(NAMES):
        if t[i] == 0 or t[i] < t[0]:
            continue
        print("  %2d  %-52s t = %7d  (+%6d shader clocks)" % (i, NAMES[i], t[i] - t[0], t[i] - prev))
        prev = t[i]
    print("  block totals of the 4 runs:", [r[21] - r[0] for r in runs])
