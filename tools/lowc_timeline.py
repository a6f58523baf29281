"""Shader-clock timeline of wave 0 of one mid-launch block of the three lowc.hip kernels (tuning build -DCP_LOWC_STAMP).
   make -C centerpose_amd/csrc variant VAR=lst FILES=lowc DEFS=-DCP_LOWC_STAMP
   CENTERPOSE_HIP_LIB=$PWD/centerpose_amd/libcenterpose_hip_lst.so python tools/lowc_timeline.py [--b 64]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from centerpose_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--b", type=int, default=64)
a = ap.parse_args()
pipe = bench.Pipeline("full", a.b, torch.device("cuda:0"), seed=317, precision="f16x3")
for _ in range(3):
    pipe.step()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
hip.lib().cp_debug_read_lowc_clk(buf)
names = ["start", "weights arrived", "activation scale arrived", "own share staged", "barrier passed", "first row done",
         "all rows done", "amax committed"]
for kind, label in ((0, "stem 7x7 3->16"), (1, "level0 3x3 16->16"), (2, "level1 3x3/2 16->32")):
    t = list(buf[kind * 16:kind * 16 + 8])
    print(label)
    for i in range(1, 8):
        if t[i]:
            print("   %-28s t = %6d  (+%6d shader clocks)" % (names[i], t[i] - t[0], t[i] - t[i - 1]))
