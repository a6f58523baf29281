"""Stand-in for the reference's pybind module ``_ext`` (DCNv2/src/vision.cpp:4-9): the forward entry
point with the identical 14-argument signature, routed to libcenterpose_hip.so.  Backward and the
PS-ROI pooling ops are training-only / unused by CenterPose and are not provided."""
from centerpose_amd import hip as _hip


def dcn_v2_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                   dilation_h, dilation_w, deformable_group):
    if not input.is_cuda:
        # dcn_v2.h:25-37 dispatches on the device; this build has the HIP path only
        raise RuntimeError("Not compiled with CPU support: centerpose_hip runs on the HIP device only")
    return _hip.dcn_v2_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w,
                               pad_h, pad_w, dilation_h, dilation_w, deformable_group)


def _training_only(*args, **kwargs):
    raise RuntimeError("centerpose_hip is an inference library: dcn_v2_backward / PS-ROI pooling are not built")


dcn_v2_backward = _training_only
dcn_v2_psroi_pooling_forward = _training_only
dcn_v2_psroi_pooling_backward = _training_only
