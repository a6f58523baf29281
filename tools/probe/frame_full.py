"""Batch-1 frames of the whole configs[2] chain (network from its hipGraph + decode + post-process + PnP), one at a time with a
device synchronisation in between -- run under rocprofv3 --kernel-trace --stats to see where a frame's microseconds go."""
import sys, time
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench

stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)  # graph capture needs a non-default stream
pipe = bench.Pipeline("full", 1, torch.device("cuda:0"), seed=317, precision="f16x3")
x1 = pipe.x[:1].contiguous()
for _ in range(5):
    pipe.step(x1, graph=True)
torch.cuda.synchronize()
ts = []
for _ in range(200):
    t0 = time.perf_counter()
    pipe.step(x1, graph=True)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print("p50 %.3f ms  p10 %.3f  p90 %.3f   detections in the frame: %d" % (ts[100], ts[20], ts[180], int(pipe.last[0].sum().item())))
