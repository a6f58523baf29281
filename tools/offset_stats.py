"""Offset statistics of the 16 DCNv2 layers of the synthetic dlav1_34 / dla_34 models (what the patch-resident DCN kernel's
halo and exception capacity have to cover).  usage: python tools/offset_stats.py [arch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from centerpose_amd import hip, synth  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "dlav1_34"
hip.set_default_precision("f16x3")
heads = synth.HEADS_POSE
sd = synth.make_state_dict(arch, heads, False)
m = hip.HipModel(arch, heads, sd, precision="f16x3")
x = synth.frames(2, seed=7).cuda()
names = []
for a, n in (("ida_0", 1), ("ida_1", 2), ("ida_2", 3)):
    for i in range(1, n + 1):
        names += ["dla_up.%s.proj_%d" % (a, i), "dla_up.%s.node_%d" % (a, i)]
for i in (1, 2):
    names += ["ida_up.proj_%d" % i, "ida_up.node_%d" % i]
for nm in names:
    try:
        _, om = m.forward(x, tap=nm + ".offmask")
    except RuntimeError as e:
        print(nm, "-", e)
        continue
    off = om[:, :18]
    a = off.abs()
    print("%-26s %4dx%-4d std %.2f  |d|>2: %.1f%%  |d|>3: %.1f%%  |d|>5: %.1f%%  max %.1f" % (
        nm, om.shape[2], om.shape[3], float(off.std()), 100 * float((a > 2).float().mean()),
        100 * float((a > 3).float().mean()), 100 * float((a > 5).float().mean()), float(a.max())))
