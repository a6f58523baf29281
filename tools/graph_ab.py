#!/usr/bin/env python
"""A/B: the batch-32 step enqueued launch by launch vs replayed from a hipGraph (no profiling events in either)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
side = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(side)
for wl, B in (("decode", 32), ("full", 64)):
    pipe = bench.Pipeline(wl, B, dev, seed=317, precision="f16x3")
    for graph in (False, True, False, True):
        for _ in range(3):
            pipe.step(pipe.x, graph=graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            pipe.step(pipe.x, graph=graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%s B=%d graph=%s: %.3f ms/step %.1f img/s" % (wl, B, graph, dt / 20 * 1e3, B * 20 / dt))
    del pipe
    torch.cuda.empty_cache()
