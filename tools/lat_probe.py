"""Batch-1 latency probe: N hipGraph replays (or eager calls) of cp_model_detect on one 512x512 frame.
Usage: python tools/lat_probe.py [--dbg FLAGS] [--eager] [--n 200] [--arch dlav1_34]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from centerpose_amd import hip, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dbg", type=int, default=0)
ap.add_argument("--eager", action="store_true")
ap.add_argument("--n", type=int, default=200)
ap.add_argument("--arch", default="dlav1_34")
a = ap.parse_args()
if a.dbg:
    hip.lib().cp_set_debug(a.dbg)
dev = torch.device("cuda:0")
heads = synth.HEADS_POSE
model = hip.HipModel(a.arch, heads, synth.make_state_dict(a.arch, heads), precision="f16x3")
x = synth.frames(1, seed=3).to(dev)
side = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(side)
for _ in range(5):
    model.detect(x, K=100, rep_mode=1, fit_gaussian=False, balance=2.0, graph=not a.eager)
torch.cuda.synchronize()
ts = []
for _ in range(a.n):
    t0 = time.perf_counter()
    model.detect(x, K=100, rep_mode=1, fit_gaussian=False, balance=2.0, graph=not a.eager)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print("dbg %d %s: p50 %.3f ms  p10 %.3f  p90 %.3f" % (a.dbg, "eager" if a.eager else "graph", ts[len(ts) // 2], ts[len(ts) // 10], ts[9 * len(ts) // 10]))
