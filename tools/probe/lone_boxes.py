"""Diagnostic: which boxes come out of run() but not run_batch() (or the reverse) on the synthetic network, and how degenerate
they are (tests/test_gpu_pose_chain.py: _degenerate)."""
import os
import pathlib
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests import test_gpu_pose_chain as T  # noqa: E402
from centerpose_amd import synth  # noqa: E402

det = T._detector(pathlib.Path(tempfile.mkdtemp()), extra=["--vis_thresh", "0.2"])
B = 64
x = torch.cat([synth.frames(8, seed=900 + i) for i in range(0, B, 8)])
outs = det.run_batch(x, [dict(T.META) for _ in range(B)])
paired = []
for b in range(B):
    single = det.run({"image": [x[b]]}, meta_inp=dict(T.META))
    left = list(outs[b]["boxes"])
    for x1 in single["boxes"]:
        d = [np.abs(np.asarray(x1[3], np.float64) - np.asarray(x2[3], np.float64)).max() for x2 in left]
        if d and min(d) < 1e-5:
            left.pop(int(np.argmin(d)))
            paired.append(T._degeneracy(x1))
        else:
            print("img %d: single-only box: %s" % (b, T._degeneracy(x1)))
    for x2 in left:
        print("img %d: batch-only box: %s" % (b, T._degeneracy(x2)))
k = sorted(paired, key=lambda m: -m["reproj_rms_px"])
print("%d paired boxes; worst reprojection of a paired box: %s" % (len(paired), k[:3]))
print("paired boxes the criterion would call degenerate: %d" % sum(T._degenerate(m) for m in paired))
