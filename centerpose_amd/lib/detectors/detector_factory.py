"""detectors/detector_factory.py:7-9"""
from .object_pose import ObjectPoseDetector

detector_factory = {
    'object_pose': ObjectPoseDetector,
}
