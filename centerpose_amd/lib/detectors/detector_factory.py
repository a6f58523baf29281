"""Task name -> detector class: the lookup ``demo.py`` does through ``detector_factory[opt.task]``
(reference detectors/detector_factory.py:7-9)."""
from . import object_pose as _object_pose

detector_factory = dict(object_pose=_object_pose.ObjectPoseDetector)
