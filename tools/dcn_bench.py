"""DCNv2 micro-benchmark: the dla_up 64->64 @128x128 layer (the heaviest of the 16, 5 of them per image) at batch 32
through the stand-alone C entry point (the layout conversions around it are separate kernels).
usage: python tools/dcn_bench.py [--dbg N] [--c 64] [--co 64] [--hw 128] [--b 32] [--n 10]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from centerpose_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dbg", type=int, default=0)
ap.add_argument("--c", type=int, default=64)
ap.add_argument("--co", type=int, default=64)
ap.add_argument("--hw", type=int, default=128)
ap.add_argument("--b", type=int, default=32)
ap.add_argument("--n", type=int, default=10)
ap.add_argument("--std", type=float, default=2.0)
a = ap.parse_args()
hip.set_default_precision("f16x3")
if a.dbg:
    hip.lib().cp_set_debug(a.dbg)
g = torch.Generator().manual_seed(1)
x = torch.randn(a.b, a.c, a.hw, a.hw, generator=g).cuda()
w = (torch.randn(a.co, a.c, 3, 3, generator=g) / (a.c * 9) ** 0.5).cuda()
bias = torch.randn(a.co, generator=g).cuda()
off = (torch.randn(a.b, 18, a.hw, a.hw, generator=g) * a.std).cuda()
mask = torch.rand(a.b, 9, a.hw, a.hw, generator=g).cuda()
for _ in range(2):
    y = hip.dcn_v2_forward(x, w, bias, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.n):
    y = hip.dcn_v2_forward(x, w, bias, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.n
print("dcn %dx%d %d->%d B%d off_std %.1f dbg %d: %.3f ms per call incl. layout kernels" % (a.hw, a.hw, a.c, a.co, a.b, a.std, a.dbg, dt * 1e3))
