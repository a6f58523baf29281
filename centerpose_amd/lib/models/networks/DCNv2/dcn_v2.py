"""Forward-only ``DCNv2`` / ``DCN`` shells over ``_ext.dcn_v2_forward`` (the HIP kernels behind include/centerpose_hip.h).

What is contractual here is the reference's surface (DCNv2/dcn_v2.py:57-128): the constructor argument order, the state-dict
names (``weight``, ``bias``, ``conv_offset_mask.weight`` / ``.bias``), the initial values, and what ``forward`` means.  The
engine (csrc/engine.hip) never goes through these modules; they exist for code that builds the reference's layers one by one
(the reference's own ``testcpu.py`` self-checks run against them, tests/test_gpu_parity.py)."""
import collections

import torch
from torch import nn

from . import _ext

_Geometry = collections.namedtuple("_Geometry", "kernel stride pad dilation groups")


def _two(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def dcn_v2_conv(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups):
    """What ``_DCNv2.apply`` computes in the forward direction (dcn_v2.py:16-35); no autograd graph is recorded."""
    (sh, sw), (ph, pw), (dh, dw) = _two(stride), _two(padding), _two(dilation)
    with torch.no_grad():
        return _ext.dcn_v2_forward(input, weight, bias, offset, mask, int(weight.shape[2]), int(weight.shape[3]),
                                   sh, sw, ph, pw, dh, dw, int(deformable_groups))


class DCNv2(nn.Module):
    """Modulated deformable convolution whose offsets and masks are inputs (dcn_v2.py:57-91)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        g = _Geometry(_two(kernel_size), _two(stride), _two(padding), _two(dilation), int(deformable_groups))
        self.in_channels, self.out_channels = in_channels, out_channels
        # the reference's attribute names, for callers that read them back
        self.kernel_size, self.stride, self.padding, self.dilation, self.deformable_groups = g
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *g.kernel))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def taps(self):
        """Sampling points per output pixel: kernel area x deformable groups (a mask value each, two offsets each)."""
        return self.deformable_groups * self.kernel_size[0] * self.kernel_size[1]

    def reset_parameters(self):
        bound = float(self.in_channels * self.kernel_size[0] * self.kernel_size[1]) ** -0.5   # 1 / sqrt(fan-in), dcn_v2.py:74-77
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.zeros_(self.bias)

    def _apply_op(self, x, offset, mask):
        return dcn_v2_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                           self.deformable_groups)

    def forward(self, input, offset, mask):
        n = self.taps()
        if offset.shape[1] != 2 * n or mask.shape[1] != n:
            raise AssertionError("DCNv2: offset / mask need %d / %d channels, got %d / %d"
                                 % (2 * n, n, offset.shape[1], mask.shape[1]))
        return self._apply_op(input, offset, mask)


class DCN(DCNv2):
    """DCNv2 that predicts its own offsets and masks with a zero-initialised convolution of the same geometry
    (dcn_v2.py:94-128): the 3n channels of that convolution are [2n offsets | n mask logits]."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        self.conv_offset_mask = nn.Conv2d(in_channels, 3 * self.taps(), self.kernel_size, self.stride, self.padding, bias=True)
        for t in (self.conv_offset_mask.weight, self.conv_offset_mask.bias):
            nn.init.zeros_(t)

    def forward(self, input):
        n = self.taps()
        om = self.conv_offset_mask(input)
        return self._apply_op(input, om[:, :2 * n].contiguous(), torch.sigmoid(om[:, 2 * n:]))
