"""100 MHz-clock timelines of workgroup (0, 0) of the igemm16p launches of one batch-1 frame (tuning build -DCP_EXP=512):
   make -C centerpose_amd/csrc variant VAR=tl FILES=igemm16 DEFS=-DCP_EXP=512
   CENTERPOSE_HIP_LIB=centerpose_amd/libcenterpose_hip_tl.so python tools/probe/small_launch_timeline.py [--dbg 4] [--arch dla_34]
Stamps 3 -> 4 and 7 -> 8 bracket an added s_waitcnt vmcnt(0) (all prologue tiles landed / slab stores drained)."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from centerpose_amd import hip, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dbg", type=int, default=0)
ap.add_argument("--arch", default="dla_34")
a = ap.parse_args()
L = hip.lib()
L.cp_set_debug(a.dbg)
heads = synth.HEADS_POSE
model = hip.HipModel(a.arch, heads, synth.make_state_dict(a.arch, heads), precision="f16x3")
x = synth.frames(1, seed=3).to("cuda")
buf = (ctypes.c_ulonglong * (64 * 16 + 1))()
for _ in range(6):
    model(x, sigmoid_hm=True)
    torch.cuda.synchronize()
assert L.cp_debug_read_tl(buf) == 0
n0 = int(buf[64 * 16])
model(x, sigmoid_hm=True)
torch.cuda.synchronize()
assert L.cp_debug_read_tl(buf) == 0
n1 = int(buf[64 * 16])
print("%s dbg %d: %d igemm16p launches per frame (eager launches; us since the workgroup's entry)" % (a.arch, a.dbg, n1 - n0))
print("%5s %5s %6s %3s %3s %2s %5s | %6s %6s %6s %6s %6s %6s %6s %6s | gap to the previous launch's end" %
      ("Cin", "Cout", "M", "sk", "n", "PD", "MHz", "setup", "issued", "scale", "landed", "LDS", "Kloop", "stores", "drain"))
prev_end = None
tot = [0.0] * 9
for k in range(n0, n1):
    r = [int(v) for v in buf[(k % 64) * 16:(k % 64) * 16 + 16]]
    d = [(r[i] - r[0]) * 0.01 for i in range(1, 9)]
    gap = (r[0] - prev_end) * 0.01 if prev_end is not None else 0.0
    prev_end = r[8]
    print("%5d %5d %6d %3d %3d %2d %5d | %s | %7.2f" % (r[9], r[10], r[11], r[12], r[13], r[14], int(r[15] / max(1e-9, (r[8] - r[0]) * 0.01)), " ".join("%6.2f" % v for v in d), gap))
    for i in range(8):
        tot[i] += d[i] - (d[i - 1] if i else 0.0)
    tot[8] += gap
print("sums over the frame's launches, us per phase:", " ".join("%.1f" % v for v in tot[:8]), " gaps (other kernels + launch) %.1f" % tot[8])
L.cp_set_debug(0)
