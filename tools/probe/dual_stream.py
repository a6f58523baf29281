"""A/B: one B=64 step on one stream vs two B=32 half-batches on two streams (do the latency-bound and the matrix-bound
kernels of the two halves overlap?).  usage: python tools/probe/dual_stream.py [--offset]"""
import sys, time
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
steps = 12

def run(pipes, streams, stagger=False):
    def one():
        for p, s in zip(pipes, streams):
            with torch.cuda.stream(s):
                p.step()
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

full = bench.Pipeline("full", 64, dev, seed=317, precision="f16x3")
ms = run([full], [torch.cuda.current_stream()])
print("one stream  B=64      : %.3f ms/step  %.1f img/s" % (ms, 64 / ms * 1e3))
del full
torch.cuda.empty_cache()
for nb in (2, 4):
    b = 64 // nb
    pipes = [bench.Pipeline("full", b, dev, seed=317 + i, precision="f16x3") for i in range(nb)]
    ms1 = run(pipes, [torch.cuda.current_stream()] * nb)
    print("one stream  %d x B=%d   : %.3f ms/step  %.1f img/s" % (nb, b, ms1, 64 / ms1 * 1e3))
    streams = [torch.cuda.Stream() for _ in range(nb)]
    ms2 = run(pipes, streams)
    print("%d streams   %d x B=%d   : %.3f ms/step  %.1f img/s" % (nb, nb, b, ms2, 64 / ms2 * 1e3))
    del pipes
    torch.cuda.empty_cache()
