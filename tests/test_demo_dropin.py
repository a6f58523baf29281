"""INTEGRATION level 0, executed: the reference's own ``src/demo.py`` (demo.py:22-87) runs unmodified against this
repo's package -- `lib.opts`, `lib.detectors.detector_factory`, `Detector(opt).run(image_path, meta_inp=meta)` and the
nine timing keys it prints.  Needs the reference tree, so it only runs in the build container (skipped elsewhere); the
device stages are replaced by the oracle there (no GPU), everything else is the product's host code."""
import os
import subprocess
import sys

import pytest
import torch

from centerpose_amd import synth

REF_SRC = "/root/reference/src"
IMG_DIR = "/root/reference/images/CenterPose/chair"


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF_SRC, "demo.py")), reason="reference tree not present")
def test_reference_demo_py_runs_unmodified_against_the_mirror(tmp_path):
    images = sorted(f for f in os.listdir(IMG_DIR) if f.endswith(".png"))
    ck = os.path.join(str(tmp_path), "dla34_synth.pth")
    torch.save({"epoch": 1, "state_dict": synth.make_state_dict("dla_34", synth.HEADS_POSE)}, ck)
    script = os.path.join(os.path.dirname(__file__), "demo_dropin_script.py")
    r = subprocess.run([sys.executable, script, REF_SRC, os.path.join(IMG_DIR, images[0]), ck], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DEMO_DROPIN_OK" in r.stdout
    line = [l for l in r.stdout.splitlines() if l.startswith("Frame 0|")]
    assert line, r.stdout[-2000:]
    for key in ("tot", "load", "pre", "net", "dec", "post", "merge", "pnp", "track"):   # demo.py:20 time_stats
        assert key + " " in line[0]
