"""Drop-in mirror of the reference's ``lib`` package for the inference hot path.

Put ``centerpose_amd/`` on ``sys.path`` (instead of the reference's ``src/``) and the reference's
``demo.py`` imports — ``lib.opts.opts``, ``lib.detectors.detector_factory.detector_factory``,
``lib.models.model.create_model/load_model``, ``lib.models.decode.object_pose_decode``,
``lib.utils.pnp.cuboid_pnp_shell.pnp_shell`` — resolve here, with the compute routed to
libcenterpose_hip.so.  Only what the inference hot path needs is mirrored: detectors (incl. the CenterPoseTrack
per-frame loop with ``lib.utils.tracker.Tracker`` / ``Tracker_baseline``), opts, model factory, decode, PnP packaging;
no training, datasets, losses, evaluation or drawing.  The reference's own ``src/demo.py`` runs unmodified against it
(tests/test_demo_dropin.py).
"""
import os as _os
import sys as _sys

# Works both as ``centerpose_amd.lib`` and as top-level ``lib`` (drop-in mode, ``centerpose_amd/`` on sys.path):
# the compute bindings are always imported by their absolute name.
_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.append(_root)
