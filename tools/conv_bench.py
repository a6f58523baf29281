"""Micro-benchmark of one convolution shape through the C ABI (for rocprofv3 counter runs).
   python tools/conv_bench.py --B 32 --H 128 --W 128 --cin 64 --cout 256 --k 3 --prec f16x3 --iters 20"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from centerpose_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--H", type=int, default=128)
ap.add_argument("--W", type=int, default=128)
ap.add_argument("--cin", type=int, default=64)
ap.add_argument("--cout", type=int, default=256)
ap.add_argument("--k", type=int, default=3)
ap.add_argument("--stride", type=int, default=1)
ap.add_argument("--prec", default="f16x3")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dbg", type=int, default=0, help="cp_set_debug flags (see engine.hip: g_dbg)")
ap.add_argument("--check", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
hip.set_default_precision(a.prec)
hip.lib().cp_set_debug(a.dbg)
x = torch.randn(a.B, a.H, a.W, a.cin, device=dev)
w = torch.randn(a.cout, a.cin, a.k, a.k, device=dev) / (a.cin * a.k * a.k) ** 0.5
for _ in range(3):
    y = hip.conv2d_nhwc(x, w, None, None, None, a.stride, a.k // 2, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    y = hip.conv2d_nhwc(x, w, None, None, None, a.stride, a.k // 2, 1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * a.cout * a.cin * a.k * a.k
if a.check:
    hip.set_default_precision("f32")
    y_f32 = hip.conv2d_nhwc(x, w, None, None, None, a.stride, a.k // 2, 1)
    torch.cuda.synchronize()
    print("check: |f16x3 - f32| max %.3e (|y| max %.2f) " % (
        float((y - y_f32).abs().max()), float(y_f32.abs().max())), end="")
print("dbg=%d " % a.dbg, end="")
print("%s B%d %dx%d %d->%d k%d: %.3f ms  %.1f TFLOP/s (incl. weight pack)" % (a.prec, a.B, a.H, a.W, a.cin, a.cout, a.k, dt * 1e3, fl / dt / 1e12))
