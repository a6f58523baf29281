"""Diagnostic: which boxes come out of run() but not run_batch() (or the reverse) on the synthetic network."""
import sys, tempfile, pathlib
import numpy as np, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests import test_gpu_pose_chain as T
from centerpose_amd import synth
det = T._detector(pathlib.Path(tempfile.mkdtemp()), extra=["--vis_thresh", "0.2"])
B = 64
x = torch.cat([synth.frames(8, seed=900 + i) for i in range(0, B, 8)])
outs = det.run_batch(x, [dict(T.META) for _ in range(B)])
for b in range(B):
    single = det.run({"image": [x[b]]}, meta_inp=dict(T.META))
    left = list(outs[b]["boxes"])
    for x1 in single["boxes"]:
        d = [np.abs(np.asarray(x1[3], np.float64) - np.asarray(x2[3], np.float64)).max() for x2 in left]
        if d and min(d) < 1e-5:
            left.pop(int(np.argmin(d)))
        else:
            print("img %d: single-only box, nearest input distance %s, score %.6f, scale %s" % (b, min(d) if d else None, x1[4]["score"], np.asarray(x1[2])))
    for x2 in left:
        d = [np.abs(np.asarray(x1[3], np.float64) - np.asarray(x2[3], np.float64)).max() for x1 in single["boxes"]]
        print("img %d: batch-only box, nearest input distance %s, score %.6f" % (b, min(d) if d else None, x2[4]["score"]))
