"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against the CPU oracle and the
committed golden vectors (outputs of the reference's own code).  Tolerances: heat-maps 1e-3 (north star),
decode bit-exact on identical head tensors, pose 1 deg / 1 %."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from centerpose_amd import hip, synth
from oracle import backbone as ob
from oracle import dcn as odcn
from oracle import decode as odec
from oracle import pnp as opnp
from oracle.tools import make_goldens as mg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CONFIGS = [("dla_34", False), ("dlav1_34", False), ("dla_34", True), ("dlav1_34", True)]


def _conv_ref(x, w, sc, sh, res, stride, pad, act):
    y = F.conv2d(x, w, None, stride, pad)
    if sc is not None:
        y = y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if res is not None:
        y = y + res
    return F.relu(y) if act == 1 else torch.sigmoid(y) if act == 2 else y


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p,act,res", [
    (1, 8, 8, 16, 16, 3, 1, 1, 0, False),      # 16x16x4 MFMA tile, ragged M
    (2, 16, 16, 4, 16, 7, 1, 3, 1, False),     # stem: Cin padded to 4, K = 196 padded to 208
    (1, 32, 32, 16, 32, 3, 2, 1, 1, False),    # N tile 32, stride 2
    (2, 16, 16, 64, 64, 3, 1, 1, 1, True),     # N tile 64 + residual (BasicBlock)
    (1, 16, 16, 64, 128, 3, 1, 1, 1, True),    # N tile 128
    (1, 8, 8, 128, 256, 1, 1, 0, 0, False),    # 1x1 (Root / project)
    (3, 12, 20, 48, 64, 3, 1, 1, 2, False),    # non power-of-two image, sigmoid
    (1, 16, 16, 64, 192, 3, 1, 1, 0, False),   # fused GRU gates
    (1, 512, 512, 16, 16, 3, 1, 1, 1, False),  # full-size level0 layer
])
def test_igemm_conv_vs_torch(device, B, H, W, Cin, Cout, k, s, p, act, res):
    g = torch.Generator().manual_seed(B * 1000 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    sc = torch.rand(Cout, generator=g) + 0.5
    sh = torch.randn(Cout, generator=g)
    y = F.conv2d(x, w, None, s, p)
    r = torch.randn(y.shape, generator=g) if res else None
    ref = _conv_ref(x, w, sc, sh, r, s, p, act)
    out = hip.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(device), w.to(device), sc.to(device), sh.to(device),
                          r.permute(0, 2, 3, 1).contiguous().to(device) if res else None, s, p, act)
    torch.testing.assert_close(out.permute(0, 3, 1, 2).cpu(), ref, rtol=0, atol=2e-5 * max(1.0, float(ref.abs().max())))


def test_dcn_reference_known_answer_and_golden(device):
    gold = np.load(os.path.join(GOLD, "dcn_ref.npz"))
    # the reference's KAT pattern (DCNv2/testcpu.py:32-67) on the fast path's channel counts: identity weights on 16 channels
    # (the KAT's own 2-channel shape runs below, test_dcn_reference_selfchecks_run_unmodified_through_ext)
    x = torch.randn(2, 16, 4, 4, generator=torch.Generator().manual_seed(1))
    w = torch.zeros(64, 16, 3, 3)
    for c in range(16):
        w[c, c, 1, 1] = 1.0
    out = hip.dcn_v2_forward(x.to(device), w.to(device), torch.zeros(64, device=device),
                             torch.zeros(2, 18, 4, 4, device=device), torch.full((2, 9, 4, 4), 0.5, device=device),
                             3, 3, 1, 1, 1, 1, 1, 1, 1).cpu()
    assert float((out[:, :16] * 2 - x).abs().max()) < 1e-10
    assert float(out[:, 16:].abs().max()) == 0.0
    # golden: output of the REFERENCE's compiled CPU im2col + GEMM on a random-offset case
    x, w, b, off, mask = mg.dcn_case()
    out = hip.dcn_v2_forward(*(t.to(device) for t in (x, w, b, off, mask)), 3, 3, 1, 1, 1, 1, 1, 1, 1).cpu()
    np.testing.assert_allclose(out.numpy(), gold["y"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("B,C,Co,H,W,std", [(2, 64, 64, 16, 16, 2.0), (1, 256, 128, 8, 8, 6.0),
                                            (1, 64, 64, 128, 128, 1.5), (2, 128, 256, 32, 32, 3.0)])
def test_dcn_vs_oracle(device, B, C, Co, H, W, std):
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(Co, generator=g)
    off = torch.randn(B, 18, H, W, generator=g) * std  # std 6 on an 8x8 map: most samples leave the image
    mask = torch.rand(B, 9, H, W, generator=g)
    ref = odcn.dcn_v2_forward(x, w, b, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    out = hip.dcn_v2_forward(*(t.to(device) for t in (x, w, b, off, mask)), 3, 3, 1, 1, 1, 1, 1, 1, 1).cpu()
    torch.testing.assert_close(out, ref, rtol=0, atol=2e-5 * float(ref.abs().max()))


def test_dcn_reference_selfchecks_run_unmodified_through_ext(device):
    """The reference's own DCNv2 self-checks through the `_ext` shim with their ORIGINAL shapes: `check_zero_offset`
    (DCNv2/testcpu.py:17-67: N, inC, outC, inH, inW = 2, 2, 2, 4, 4, identity weights, zero offsets, mask 0.5 ->
    `(input - 2 * output).abs().max() < 1e-10`) and `example_dconv` (:169-180: DCN(64, 64, 3x3, deformable_groups=2)
    on [2, 64, 128, 128]) -- neither fits the CenterPose-only fast path (C % 16 / dg 1); the generic kernel takes them."""
    from centerpose_amd.lib.models.networks.DCNv2.dcn_v2 import DCN, DCNv2

    g = torch.Generator().manual_seed(0)
    N, inC, outC, inH, inW, dg = 2, 2, 2, 4, 4, 1
    dcn = DCNv2(inC, outC, (3, 3), stride=1, padding=1, dilation=1, deformable_groups=dg).to(device)
    with torch.no_grad():   # conv_identify (testcpu.py:23-30)
        dcn.weight.zero_()
        dcn.bias.zero_()
        for c in range(inC):
            dcn.weight[c, c, 1, 1] = 1.0
    x = torch.randn(N, inC, inH, inW, generator=g)
    offset = torch.zeros(N, dg * 18, inH, inW)          # zeroed conv_offset (weights and bias 0)
    mask = torch.sigmoid(torch.zeros(N, dg * 9, inH, inW))
    out = dcn(x.to(device), offset.to(device), mask.to(device)).cpu()
    assert float((x - out * 2).abs().max()) < 1e-10
    # example_dconv: offsets / masks come from DCN's own conv_offset_mask (given non-zero weights here so that the
    # deformable groups really differ), reference = the oracle on the same offsets
    dcn2 = DCN(64, 64, kernel_size=(3, 3), stride=1, padding=1, deformable_groups=2)
    with torch.no_grad():
        dcn2.conv_offset_mask.weight.copy_(torch.randn(dcn2.conv_offset_mask.weight.shape, generator=g) * 0.05)
        dcn2.conv_offset_mask.bias.copy_(torch.randn(54, generator=g) * 0.5)
        dcn2.bias.copy_(torch.randn(64, generator=g))
    x = torch.randn(2, 64, 128, 128, generator=g)
    with torch.no_grad():
        om = dcn2.conv_offset_mask(x)
        o1, o2, m = torch.chunk(om, 3, dim=1)
        off, m = torch.cat((o1, o2), dim=1), torch.sigmoid(m)
        ref = odcn.dcn_v2_forward(x, dcn2.weight, dcn2.bias, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 2)
        out = dcn2.to(device)(x.to(device)).cpu()
    assert out.shape == (2, 64, 128, 128)
    torch.testing.assert_close(out, ref, rtol=0, atol=3e-5 * float(ref.abs().max()))


@pytest.mark.parametrize("i", range(len(mg.DCN_GENERIC_SHAPES)))
def test_dcn_generic_shapes_vs_reference_golden(device, i):
    """Everything `_ext.dcn_v2_forward` accepts beyond CenterPose's own use (dcn_v2.h:9-23) against outputs of the
    reference's compiled CPU source (tests/golden/dcn_generic_ref.npz) and the oracle port."""
    gold = np.load(os.path.join(GOLD, "dcn_generic_ref.npz"))["y%d" % i]
    x, w, b, off, mask, args = mg.dcn_generic_case(i)
    ref = odcn.dcn_v2_forward(x, w, b, off, mask, *args)
    out = hip.dcn_v2_forward(*(t.to(device) for t in (x, w, b, off, mask)), *args).cpu()
    assert out.shape == ref.shape == gold.shape
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    torch.testing.assert_close(out, ref, rtol=0, atol=tol)
    np.testing.assert_allclose(out.numpy(), gold, rtol=0, atol=tol)


def test_dcn_generic_and_fast_path_agree(device):
    """The same CenterPose-shaped layer through both kernels (cp_set_debug 8388608 forces the generic one)."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 64, 32, 32, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(64, generator=g)
    off = torch.randn(2, 18, 32, 32, generator=g) * 2.0
    mask = torch.rand(2, 9, 32, 32, generator=g)
    t = [v.to(device) for v in (x, w, b, off, mask)]
    fast = hip.dcn_v2_forward(*t, 3, 3, 1, 1, 1, 1, 1, 1, 1).cpu()
    hip.lib().cp_set_debug(8388608)
    try:
        slow = hip.dcn_v2_forward(*t, 3, 3, 1, 1, 1, 1, 1, 1, 1).cpu()
    finally:
        hip.lib().cp_set_debug(0)
    torch.testing.assert_close(fast, slow, rtol=0, atol=2e-5 * float(slow.abs().max()))


def test_dcn_rejects_inconsistent_arguments(device):
    x = torch.zeros(1, 6, 4, 4, device=device)
    with pytest.raises(RuntimeError):   # C not divisible by deformable_group
        hip.dcn_v2_forward(x, torch.zeros(4, 6, 3, 3, device=device), torch.zeros(4, device=device),
                           torch.zeros(1, 72, 4, 4, device=device), torch.zeros(1, 36, 4, 4, device=device),
                           3, 3, 1, 1, 1, 1, 1, 1, 4)
    with pytest.raises(RuntimeError):   # offset tensor of the wrong group count (dcn_v2_cuda.cu:60-66)
        hip.dcn_v2_forward(x, torch.zeros(4, 6, 3, 3, device=device), torch.zeros(4, device=device),
                           torch.zeros(1, 18, 4, 4, device=device), torch.zeros(1, 18, 4, 4, device=device),
                           3, 3, 1, 1, 1, 1, 1, 1, 2)


@pytest.fixture(params=["f32", "f16x3"])
def precision(request):
    """Both arithmetic modes of the contraction kernels (centerpose_hip.h: CP_PREC_*)."""
    hip.set_default_precision(request.param)
    yield request.param
    hip.set_default_precision("f32")


@pytest.mark.parametrize("Cin,Cout,k,s,res", [(32, 64, 3, 2, False), (64, 256, 3, 1, True), (128, 128, 3, 1, True),
                                              (448, 128, 1, 1, False), (64, 27, 3, 1, False), (64, 192, 3, 1, False)])
def test_conv_both_precisions_vs_float64(device, precision, Cin, Cout, k, s, res):
    """Split-binary16 products must stay float32-class: error vs a float64 convolution <= 2e-5 of the output range
    (observed ~3e-6 for f32, ~6e-6 for f16x3)."""
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(2, Cin, 20, 24, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    y = F.conv2d(x.double(), w.double(), None, s, k // 2)
    r = torch.randn(y.shape, generator=g) if res else None
    ref = (y + r.double()) if res else y
    out = hip.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(device), w.to(device), None, None,
                          r.permute(0, 2, 3, 1).contiguous().to(device) if res else None, s, k // 2, 0)
    err = float((out.permute(0, 3, 1, 2).cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err


@pytest.mark.parametrize("B,H,W,Cin,Cout,res,act", [
    (2, 16, 16, 32, 64, False, 0),     # two K steps, N tile 64
    (1, 20, 24, 448, 128, True, 1),    # M % 128 != 0 (ragged last tile), residual + ReLU, N tile 128
    (3, 8, 8, 1280, 512, False, 1),    # the deepest Root node's shape: 80 K steps, four N tiles
    (1, 12, 20, 64, 256, True, 0),
])
def test_pointwise_stream_kernel_vs_float64_and_lds_loop(device, f16x3, B, H, W, Cin, Cout, res, act):
    """pw16.hip (1x1 layers as a register-only stream: A fragments straight from global memory, weight fragments from the
    fragment-ordered copy) against a float64 convolution and against the LDS-staged loop it replaces (cp_set_debug
    4194304): the same products summed in the same order."""
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    y = F.conv2d(x.double(), w.double())
    r = torch.randn(y.shape, generator=g) if res else None
    ref = y + r.double() if res else y
    if act:
        ref = ref.clamp(min=0)
    args = (x.permute(0, 2, 3, 1).contiguous().to(device), w.to(device), None, None,
            r.permute(0, 2, 3, 1).contiguous().to(device) if res else None, 1, 0, act)
    out = hip.conv2d_nhwc(*args).permute(0, 3, 1, 2).cpu()
    hip.lib().cp_set_debug(4194304)
    try:
        old = hip.conv2d_nhwc(*args).permute(0, 3, 1, 2).cpu()
    finally:
        hip.lib().cp_set_debug(0)
    hip.lib().cp_set_debug(4)   # round 5's form of the stream: fragment-shaped A loads instead of whole lines through staging rows
    try:
        frag = hip.conv2d_nhwc(*args).permute(0, 3, 1, 2).cpu()
    finally:
        hip.lib().cp_set_debug(0)
    scale = float(ref.abs().max())
    assert float((out.double() - ref).abs().max()) < 2e-5 * scale
    assert float((out - old).abs().max()) < 2e-6 * scale
    assert torch.equal(out, frag)   # the same products in the same order: only the way the A operand reaches the registers differs


@pytest.mark.parametrize("B,H,W,Cin,Cout,res,act", [
    (2, 16, 32, 64, 128, False, 1),     # N tile 128, one 64-channel chunk
    (1, 24, 16, 128, 64, True, 1),      # N tile 64, two chunks, residual (BasicBlock conv2)
    (3, 8, 48, 64, 27, False, 0),       # N tile 32 (conv_offset_mask shape)
    (1, 32, 32, 256, 256, True, 0),     # four chunks, two N tiles
    (2, 16, 16, 64, 192, False, 0),     # ConvGRU input side (3 x 64)
])
def test_halo_resident_conv3x3_vs_float64_and_previous_kernel(device, f16x3, B, H, W, Cin, Cout, res, act):
    """halo16.hip (3x3 / stride 1 layers whose maps tile into 8x16 patches) against a float64 convolution, and against
    the per-tap implicit-GEMM kernel it replaces (cp_set_debug 4096): same products, different summation order."""
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    y = F.conv2d(x.double(), w.double(), None, 1, 1)
    r = torch.randn(y.shape, generator=g) if res else None
    ref = (y + r.double()) if res else y
    ref = F.relu(ref) if act == 1 else ref
    args = (x.permute(0, 2, 3, 1).contiguous().to(device), w.to(device), None, None,
            r.permute(0, 2, 3, 1).contiguous().to(device) if res else None, 1, 1, act)
    hip.lib().cp_set_debug(8192)   # the halo kernel for every eligible N tile (default: the 32-wide one only)
    try:
        out = hip.conv2d_nhwc(*args).permute(0, 3, 1, 2).cpu().double()
    finally:
        hip.lib().cp_set_debug(0)
    hip.lib().cp_set_debug(4096)
    try:
        old = hip.conv2d_nhwc(*args).permute(0, 3, 1, 2).cpu().double()
    finally:
        hip.lib().cp_set_debug(0)
    assert float((out - ref).abs().max() / ref.abs().max()) < 2e-5
    assert float((out - old).abs().max() / ref.abs().max()) < 2e-6
    assert not torch.equal(out, old) or Cin == 64   # two different kernels really ran (summation order differs)


@pytest.mark.parametrize("B,H,W,act", [
    (2, 16, 32, 0),     # one strip, one band per image
    (3, 8, 48, 1),      # a ragged second strip (16 of 32 columns), a band of 8 rows, ReLU
    (1, 40, 40, 2),     # ragged strip and ragged last band (8 of 16 rows), sigmoid on every channel
    (2, 33, 100, 0),    # odd height: a band of ONE row (an odd row count runs one more turn of the two-row loop); 4 strips
    (5, 128, 128, 2),   # the offset convolutions' own map size: more jobs than one workgroup's waves, several jobs per wave
])
def test_streamed_conv3x3_c64_n32_vs_float64_and_other_kernels(device, f16x3, B, H, W, act):
    """strm16.hip (64 -> <= 32 channel 3x3 layers as wave-private row streams, weights in LDS; cp_set_debug 536870912: at any size)
    against a float64 convolution and against the kernel it replaces (268435456: halo16's 32-wide tile where the map tiles into
    8 x 16 patches, the per-tap implicit GEMM otherwise -- both sum in the same order); the sigmoid by hardware exp2 / rcp
    (|error| < 3e-7) instead of expf.  How the profile shows that strm16 ran: test_backbone_at_bench_batch_every_image_every_launch."""
    g = torch.Generator().manual_seed(B * 100 + H + W)
    x = torch.randn(B, 64, H, W, generator=g)
    w = torch.randn(32, 64, 3, 3, generator=g) / (64 * 9) ** 0.5
    scale = 0.5 + torch.rand(32, generator=g)
    shift = torch.randn(32, generator=g)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    ref = F.relu(ref) if act == 1 else torch.sigmoid(ref) if act == 2 else ref
    args = (x.permute(0, 2, 3, 1).contiguous().to(device), w.to(device), scale.to(device), shift.to(device), None, 1, 1, act)
    outs = {}
    for name, dbg in (("strm16", 536870912), ("other", 268435456)):
        hip.lib().cp_set_debug(dbg)
        try:
            outs[name] = hip.conv2d_nhwc(*args).permute(0, 3, 1, 2).cpu().double()
            again = hip.conv2d_nhwc(*args).permute(0, 3, 1, 2).cpu().double()
        finally:
            hip.lib().cp_set_debug(0)
        assert torch.equal(outs[name], again), name
        assert float((outs[name] - ref).abs().max() / ref.abs().max()) < 2e-5, name
    assert float((outs["strm16"] - outs["other"]).abs().max() / ref.abs().max()) < 2e-6
    if act != 2:
        # all three kernels sum a 64-channel layer in the same order (tap, 16-channel group; lo.hi, hi.lo, hi.hi): bit-equal
        assert torch.equal(outs["strm16"], outs["other"])
    else:
        assert not torch.equal(outs["strm16"], outs["other"])   # (the sigmoid forms differ: two different kernels really ran)


def test_dcn_both_precisions_vs_oracle(device, precision):
    g = torch.Generator().manual_seed(77)
    x = torch.randn(2, 64, 24, 20, generator=g)
    w = torch.randn(128, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(128, generator=g)
    off = torch.randn(2, 18, 24, 20, generator=g) * 2.5
    mask = torch.rand(2, 9, 24, 20, generator=g)
    ref = odcn.dcn_v2_forward_f64(x, w, b, off, mask)
    out = hip.dcn_v2_forward(*(t.to(device) for t in (x, w, b, off, mask)), 3, 3, 1, 1, 1, 1, 1, 1, 1).cpu()
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 2e-5


@pytest.mark.parametrize("B,C,Co,H,W,std", [
    (2, 32, 64, 8, 16, 0.0),       # one patch per image, zero offsets: every sample inside the halo, image border = zero fill
    (2, 64, 64, 16, 16, 1.5),      # the bench's offset scale: a few dozen exception samples per block
    (1, 128, 256, 32, 32, 3.0),    # four chunks, four N tiles; some blocks over the exception capacity
    (1, 64, 128, 16, 32, 8.0),     # offsets far beyond the halo and the image
    (1, 32, 64, 64, 64, 10.0),     # ... inside a larger image: every block over the exception capacity (buffer-load mode)
    (2, 64, 64, 32, 32, 2.0),      # the upper end of the synthetic network's offset scale: ~100 exception samples per block
    (1, 64, 64, 128, 128, 1.5),    # the heaviest layer shape of the network (dla_up 64 -> 64 at 128 x 128)
])
def test_dcn_patch_resident_vs_oracle_and_gather_kernel(device, B, C, Co, H, W, std):
    """dcn16p.hip (samples gathered from an LDS-staged halo, exception samples from spare patch rows, overflowing blocks
    through buffer loads) against the float64 oracle and against dcn16.hip (cp_set_debug 32768): same products, different
    summation order.  cp_set_debug 65536 selects it for launches of any size."""
    hip.set_default_precision("f16x3")
    try:
        g = torch.Generator().manual_seed(C + H + int(std * 10))
        x = torch.randn(B, C, H, W, generator=g)
        w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
        b = torch.randn(Co, generator=g)
        off = torch.randn(B, 18, H, W, generator=g) * std
        mask = torch.rand(B, 9, H, W, generator=g)
        ref = odcn.dcn_v2_forward_f64(x, w, b, off, mask)
        args = [t.to(device) for t in (x, w, b, off, mask)] + [3, 3, 1, 1, 1, 1, 1, 1, 1]
        hip.lib().cp_set_debug(65536)
        try:
            out = hip.dcn_v2_forward(*args).cpu()
        finally:
            hip.lib().cp_set_debug(0)
        hip.lib().cp_set_debug(32768)
        try:
            old = hip.dcn_v2_forward(*args).cpu()
        finally:
            hip.lib().cp_set_debug(0)
        hip.lib().cp_set_debug(65536 | 524288)   # never the 128-wide N tile
        try:
            narrow = hip.dcn_v2_forward(*args).cpu()
        finally:
            hip.lib().cp_set_debug(0)
    finally:
        hip.set_default_precision("f32")
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 2e-5
    assert float((out - old).abs().max() / ref.abs().max()) < 2e-6
    assert C == 32 or not torch.equal(out, old)   # two different kernels really ran (one 32-channel chunk: same order)
    # layers with whole 128-channel tiles take the 128-wide N tile (gather / blend / split once per 128 outputs, round 5): every
    # output sums the same products in the same order as on the 64-wide tile
    assert torch.equal(out, narrow)


@pytest.mark.parametrize("B,C,Co,H,W,std,grid8", [
    (2, 32, 64, 8, 16, 0.0, False),     # one patch per image, zero offsets, image border = zero fill by the DMA
    (2, 64, 64, 16, 16, 1.5, False),    # a few exception samples per patch
    (2, 64, 64, 32, 32, 1.5, True),     # 16 items on 8 workgroups: item -> item hand-over (prefetched chunk 0, record, weights)
    (1, 128, 256, 32, 32, 3.0, True),   # eight chunks, four N tiles (tn changes from item to item), patches over the capacity
    (1, 64, 128, 16, 32, 8.0, True),    # offsets far beyond the halo and the image: buffer-load mode, mixed with fast patches
    (1, 32, 64, 64, 64, 10.0, True),    # every patch in buffer-load mode, several per workgroup
    (2, 64, 64, 32, 32, 2.0, True),     # the upper end of the synthetic network's offset scale
    (3, 64, 64, 40, 48, 1.0, True),     # 45 items: uneven item counts per XCD, non-power-of-two map
    (1, 64, 64, 128, 128, 1.5, False),  # the heaviest layer shape of the network (dla_up 64 -> 64 at 128 x 128)
])
def test_dcn_streamed_persistent_vs_oracle_and_gather_kernel(device, B, C, Co, H, W, std, grid8):
    """dcn16s.hip (persistent workgroups, halo chunks and exception corners streamed by LDS-DMA into two buffers) against the
    float64 oracle and against dcn16.hip (cp_set_debug 32768).  cp_set_debug 65536 | 2097152 selects it for launches of any size,
    8388608 limits the grid to 8 workgroups so that small problems exercise the item-to-item hand-over."""
    hip.set_default_precision("f16x3")
    try:
        g = torch.Generator().manual_seed(C + H + int(std * 10))
        x = torch.randn(B, C, H, W, generator=g)
        w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
        b = torch.randn(Co, generator=g)
        off = torch.randn(B, 18, H, W, generator=g) * std
        mask = torch.rand(B, 9, H, W, generator=g)
        ref = odcn.dcn_v2_forward_f64(x, w, b, off, mask)
        args = [t.to(device) for t in (x, w, b, off, mask)] + [3, 3, 1, 1, 1, 1, 1, 1, 1]
        hip.lib().cp_set_debug(65536 | 2097152 | (8388608 if grid8 else 0))
        try:
            out = hip.dcn_v2_forward(*args).cpu()
            out2 = hip.dcn_v2_forward(*args).cpu()
        finally:
            hip.lib().cp_set_debug(0)
        hip.lib().cp_set_debug(32768)
        try:
            old = hip.dcn_v2_forward(*args).cpu()
        finally:
            hip.lib().cp_set_debug(0)
    finally:
        hip.set_default_precision("f32")
    assert torch.equal(out, out2)   # deterministic (no atomics on the data path, no race between DMA and gather)
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 2e-5
    assert float((out - old).abs().max() / ref.abs().max()) < 2e-6
    assert not torch.equal(out, old)   # two different kernels really ran (summation order differs)


@pytest.mark.parametrize("B,C,Co,H,W,std", [
    (2, 32, 64, 8, 16, 0.0),       # one patch per image, zero offsets: image border = zero fill, two chunks
    (2, 64, 64, 16, 16, 1.5),      # the bench's offset scale: a few dozen exception samples per block
    (1, 128, 256, 32, 32, 3.0),    # eight chunks, four N tiles; some blocks over the exception capacity
    (1, 64, 128, 16, 32, 8.0),     # offsets far beyond the halo and the image
    (1, 32, 64, 64, 64, 10.0),     # every block over the exception capacity (buffer-load mode)
    (2, 64, 64, 32, 32, 2.0),      # ~100 exception samples per block
    (3, 64, 64, 40, 48, 1.0),      # non-power-of-two map
    (1, 64, 60, 16, 32, 1.5),      # Cout not a multiple of the N tile: padded channels are not stored
    (1, 64, 64, 128, 128, 1.5),    # the heaviest layer shape of the network (dla_up 64 -> 64 at 128 x 128)
])
def test_dcn_three_workgroups_per_cu_vs_oracle_and_other_kernels(device, B, C, Co, H, W, std):
    """dcn16t.hip (dcn16p's gather on a 168-register / 46 KB budget: three workgroups per CU, 16-channel chunks, transposed product,
    first chunk requested in the prologue) against the float64 oracle, against dcn16.hip (cp_set_debug 32768: another summation
    order) and against dcn16s.hip (the same K order (16-channel chunk, tap): bit-identical).  cp_set_debug 65536 | 33554432 selects
    it for launches of any size."""
    hip.set_default_precision("f16x3")
    try:
        g = torch.Generator().manual_seed(C + H + int(std * 10))
        x = torch.randn(B, C, H, W, generator=g)
        w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
        b = torch.randn(Co, generator=g)
        off = torch.randn(B, 18, H, W, generator=g) * std
        mask = torch.rand(B, 9, H, W, generator=g)
        ref = odcn.dcn_v2_forward_f64(x, w, b, off, mask)
        args = [t.to(device) for t in (x, w, b, off, mask)] + [3, 3, 1, 1, 1, 1, 1, 1, 1]
        outs = {}
        for name, dbg in (("t", 65536 | 33554432), ("t2", 65536 | 33554432), ("gather", 32768), ("s", 65536 | 2097152 | 67108864)):
            hip.lib().cp_set_debug(dbg)
            try:
                outs[name] = hip.dcn_v2_forward(*args).cpu()
            finally:
                hip.lib().cp_set_debug(0)
    finally:
        hip.set_default_precision("f32")
    assert torch.equal(outs["t"], outs["t2"])   # deterministic
    assert float((outs["t"].double() - ref).abs().max() / ref.abs().max()) < 2e-5
    assert float((outs["t"] - outs["gather"]).abs().max() / ref.abs().max()) < 2e-6
    assert not torch.equal(outs["t"], outs["gather"])   # two different kernels really ran
    assert torch.equal(outs["t"], outs["s"])    # same products in the same order as the streamed kernel


@pytest.mark.parametrize("C,Co,HW", [(64, 64, 128), (128, 128, 64)])
def test_dcn_at_bench_batch_size_independent_properties(device, C, Co, HW):
    """The two heaviest DCNv2 shapes of the benchmark, at the benchmark's batch (B = 64: 8192 / 4096 patches, where no CPU oracle
    finishes in seconds) through properties that do not depend on the size:
      * the launch really goes to the kernel the dispatcher means -- dcn16t (three workgroups per CU, round 6) for 64 -> 64, dcn16p on
        the 128-wide N tile for 128 -> 128 -- and it agrees with another patch kernel (dcn16p: 67108864 | 1048576, resp. dcn16s:
        65536 | 2097152) and with the gather kernel dcn16 (32768) to summation-order round-off; the 128-wide tile equals the 64-wide
        one (524288 | 1048576 | 67108864) and dcn16t equals dcn16s (same K order) bit for bit;
      * homogeneity: without a bias f(4 x) == 4 f(x) bit for bit -- every operand is pre-scaled by exact powers of two
        (profiles/NOTES.md 3.1), so a power-of-two input scale must come out as exactly that scale;
      * additivity in the input for fixed offsets / masks: f(x1 + x2) - f(x1) - f(x2) + f(0) == 0 to round-off;
      * an all-zero mask gives the bias, whatever the offsets."""
    hip.set_default_precision("f16x3")
    try:
        g = torch.Generator().manual_seed(C + HW)
        B = 64
        x1 = torch.randn(B, C, HW, HW, generator=g).to(device)
        x2 = torch.randn(B, C, HW, HW, generator=g).to(device)
        w = (torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5).to(device)
        b = torch.randn(Co, generator=g).to(device)
        off = (torch.randn(B, 18, HW, HW, generator=g) * 1.5).to(device)
        mask = torch.rand(B, 9, HW, HW, generator=g).to(device)
        tail = [3, 3, 1, 1, 1, 1, 1, 1, 1]
        f = lambda x, m=mask: hip.dcn_v2_forward(x, w, b, off, m, *tail)
        y1 = f(x1)
        scale = float(y1.abs().max())
        for dbg in ((67108864 | 1048576 if Co % 128 else 65536 | 2097152), 32768):
            hip.lib().cp_set_debug(dbg)
            try:
                other = f(x1)
            finally:
                hip.lib().cp_set_debug(0)
            assert not torch.equal(other, y1), dbg            # a different kernel ran
            assert float((other - y1).abs().max()) / scale < 2e-6, dbg
        hip.lib().cp_set_debug(524288 | 1048576 | 67108864 if Co % 128 == 0 else 67108864 | 65536 | 2097152)
        try:
            assert torch.equal(f(x1), y1)
        finally:
            hip.lib().cp_set_debug(0)
        bias = b.view(1, Co, 1, 1)
        # (without a bias the property is exact: the products, their sums and the epilogue's power-of-two scales carry the factor 4
        # through unchanged; with one, fl(4 X + b) - b and 4 (fl(X + b) - b) differ by the roundings of the additions: a few ulp)
        f0 = lambda x: hip.dcn_v2_forward(x, w, torch.zeros_like(b), off, mask, *tail)
        assert torch.equal(f0(4.0 * x1), 4.0 * f0(x1))
        y4 = f(4.0 * x1)
        assert float(((y4 - bias) - 4.0 * (y1 - bias)).abs().max()) / scale < 1e-6
        y0, y2, y12 = f(torch.zeros_like(x1)), f(x2), f(x1 + x2)
        assert float((y0 - bias).abs().max()) == 0.0
        assert float((y12 - y1 - y2 + y0).abs().max()) / scale < 2e-5
        yz = f(x1, torch.zeros_like(mask))
        assert float((yz - bias).abs().max()) == 0.0
    finally:
        hip.set_default_precision("f32")


@pytest.mark.parametrize("B,C,Co,HW,n", [(16, 64, 64, 128, 300), (64, 256, 256, 32, 500)])
@pytest.mark.parametrize("kernel,dbg", [("dcn16p", 65536 | 1048576 | 67108864), ("dcn16t", 65536 | 33554432), ("dcn16p on the 64-wide N tile only", 65536 | 1048576 | 524288 | 67108864),
                                        ("dcn16s", 65536 | 2097152)])
def test_dcn_kernels_are_stable_over_many_launches(device, kernel, dbg, B, C, Co, HW, n):
    """Regression for the wrong set-up values found in round 4 and explained in round 5 (profiles/NOTES.md: a packed-f32 op with a
    set op_sel bit, which the SLP vectorizer made of the set-up's two sums in the early-prologue build, is computed wrongly in
    lanes 48-63 beside another wave's MFMAs; that build failed in 44 % of the launches of the second shape below) -- only visible
    in launches with more workgroups than the chip holds at once and never in the same place twice, so no single-launch parity
    test ever saw it.  300 launches of the heaviest layer shape at 16 images (2048 patches), and 500 of the shape and batch it was
    found on (256 -> 256 @32 x 32, B = 64: 2048 workgroups, four rounds of the chip), must all equal the gather kernel's result
    to summation-order round-off.  (Since round 5 that shape takes dcn16p's 128-wide N tile by default: both tiles are run.)"""
    hip.set_default_precision("f16x3")
    try:
        g = torch.Generator().manual_seed(5)
        x = torch.randn(B, C, HW, HW, generator=g).to(device)
        w = (torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5).to(device)
        b = torch.randn(Co, generator=g).to(device)
        off = (torch.randn(B, 18, HW, HW, generator=g) * 1.5).to(device)
        mask = torch.rand(B, 9, HW, HW, generator=g).to(device)
        args = [x, w, b, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1]
        hip.lib().cp_set_debug(32768)
        ref = hip.dcn_v2_forward(*args)
        hip.lib().cp_set_debug(dbg)
        scale = float(ref.abs().max())
        first, worst, nbad = None, 0.0, 0
        for _ in range(n):
            y = hip.dcn_v2_forward(*args)
            if first is None:
                first = y.clone()
                assert not torch.equal(first, ref)   # another kernel than the reference's really ran
            err = float((y - ref).abs().max()) / scale
            worst = max(worst, err)
            nbad += int(not torch.equal(y, first))
    finally:
        hip.lib().cp_set_debug(0)
        hip.set_default_precision("f32")
    assert worst < 2e-6, (kernel, worst)
    assert nbad == 0, (kernel, nbad)   # bit-identical from launch to launch


@pytest.fixture
def f16x3():
    hip.set_default_precision("f16x3")
    yield
    hip.set_default_precision("f32")


@pytest.mark.parametrize("xmag", [1e-5, 1e-3, 1e3, 1e5])
@pytest.mark.parametrize("wmag", [1e-4, 1e-2])
@pytest.mark.parametrize("k,Cout", [(3, 64), (1, 128), (3, 27)])
def test_f16x3_conv_is_range_safe(device, f16x3, xmag, wmag, k, Cout):
    """The split-binary16 mode must not depend on operand magnitude (binary16 has 5 exponent bits; float32, the
    reference's arithmetic -- dcn_v2_cuda.cu:58 -- has 8): activations down to 1e-5 (hi/lo halves would be subnormal)
    and up to 1e5 (> 65504 would saturate), weights down to 1e-4.  Operands are pre-scaled by exact powers of two
    (per-tensor |max| for activations, per output channel for weights), so the error stays <= 2e-5 of the range."""
    g = torch.Generator().manual_seed(int(k * 100 + Cout))
    x = torch.randn(2, 64, 20, 24, generator=g) * xmag
    w = torch.randn(Cout, 64, k, k, generator=g) * wmag
    ref = F.conv2d(x.double(), w.double(), None, 1, k // 2)
    out = hip.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(device), w.to(device), None, None, None, 1, k // 2, 0)
    err = float((out.permute(0, 3, 1, 2).cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err


def test_f16x3_conv_wide_dynamic_range_inside_one_tensor(device, f16x3):
    """A few pixels 2^12 times larger than the rest (one per-tensor scale must serve both): error relative to the
    range <= 2e-5, and the quiet region keeps float32-class relative accuracy (its |max| is 2^-12 of the tensor's)."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 64, 32, 32, generator=g)
    x[:, :, :4, :4] *= 4096.0
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    out = hip.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(device), w.to(device), None, None, None, 1, 1, 0)
    o = out.permute(0, 3, 1, 2).cpu().double()
    assert float((o - ref).abs().max() / ref.abs().max()) < 2e-5
    quiet = (slice(None), slice(None), slice(8, None), slice(8, None))
    assert float((o[quiet] - ref[quiet]).abs().max() / ref[quiet].abs().max()) < 2e-5


@pytest.mark.parametrize("xmag", [1e-5, 1e-3, 1e3, 1e5])
@pytest.mark.parametrize("wmag", [1e-4, 1e-2])
def test_f16x3_dcn_is_range_safe(device, f16x3, xmag, wmag):
    g = torch.Generator().manual_seed(78)
    x = torch.randn(2, 64, 24, 20, generator=g) * xmag
    w = torch.randn(128, 64, 3, 3, generator=g) * wmag
    b = torch.randn(128, generator=g) * xmag * wmag
    off = torch.randn(2, 18, 24, 20, generator=g) * 2.5
    mask = torch.rand(2, 9, 24, 20, generator=g)
    ref = odcn.dcn_v2_forward_f64(x, w, b, off, mask)
    out = hip.dcn_v2_forward(*(t.to(device) for t in (x, w, b, off, mask)), 3, 3, 1, 1, 1, 1, 1, 1, 1).cpu()
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 2e-5


def _rescale_bn_convs(sd, f=2.0 ** -8):
    """Every convolution that feeds an eval-mode BatchNorm gets its weight (and bias) multiplied by f and the BatchNorm
    compensated (running_mean * f, running_var' = (var + eps) * f^2 - eps), so the network computes the same function
    with conv weights / pre-normalisation activations 256 times smaller -- what trained checkpoints look like next to
    the unit-variance synthetic ones."""
    out = {k: v.clone() for k, v in sd.items()}
    n = 0
    for k in sd:
        if not k.endswith(".running_var"):
            continue
        bn = k[: -len(".running_var")]
        # conv feeding this BatchNorm: `<p>.bn1` <- `<p>.conv1`, `<p>.bn` <- `<p>.conv`, `<p>.1` <- `<p>.0`,
        # `<p>.actf.0` <- `<p>.conv` (DCN: weight + bias)
        if bn.endswith(".actf.0"):
            conv = bn[: -len(".actf.0")] + ".conv"
        elif bn.rsplit(".", 1)[1].startswith("bn"):
            conv = bn.rsplit(".", 1)[0] + ".conv" + bn.rsplit(".", 1)[1][2:]
        else:
            conv = bn.rsplit(".", 1)[0] + "." + str(int(bn.rsplit(".", 1)[1]) - 1)
        assert conv + ".weight" in sd, (bn, conv)
        out[conv + ".weight"] = sd[conv + ".weight"] * f
        if conv + ".bias" in sd:
            out[conv + ".bias"] = sd[conv + ".bias"] * f
        out[bn + ".running_mean"] = sd[bn + ".running_mean"] * f
        out[k] = (sd[k] + 1e-5) * (f * f) - 1e-5
        assert float(out[k].min()) > 0
        n += 1
    assert n > 40
    return out


@pytest.mark.parametrize("arch", ["dla_34", "dlav1_34"])
def test_backbone_small_weights_f16x3_vs_oracle(device, arch):
    """Backbone with every BatchNorm'd convolution's weights x 2^-8 (BatchNorm compensated): the f16x3 mode against the
    float32 oracle run on the SAME rescaled weights, at the heat-map gate, plus the exact-f32 mode as a control."""
    heads = synth.HEADS_POSE
    sd = _rescale_bn_convs(synth.make_state_dict(arch, heads))
    x = synth.frames(2, seed=53, h=128, w=128)
    zo = ob.dlaseg_forward(sd, x, heads, arch=arch.split("_")[0])
    zo0 = ob.dlaseg_forward(synth.make_state_dict(arch, heads), x, heads, arch=arch.split("_")[0])
    assert float((torch.sigmoid(zo["hm"]) - torch.sigmoid(zo0["hm"])).abs().max()) < 1e-3   # same function
    for prec in ("f16x3", "f32"):
        z = hip.HipModel(arch, heads, sd, precision=prec)(x.to(device), sigmoid_hm=True)
        assert float((z["hm"].cpu() - torch.sigmoid(zo["hm"])).abs().max()) < 1e-3, prec
        assert float((z["hm_hp"].cpu() - torch.sigmoid(zo["hm_hp"])).abs().max()) < 1e-3, prec
        for k in ("wh", "hps", "reg", "hp_offset", "scale"):
            assert float((z[k].cpu() - zo[k]).abs().max()) < 1e-3 * max(1.0, float(zo[k].abs().max())), (prec, k)


@pytest.mark.parametrize("arch,B,pick", [("dlav1_34", 32, 21), ("dla_34", 64, 45)])
def test_backbone_at_bench_batch_spot_parity(device, arch, B, pick):
    """BASELINE configs[1] / configs[2] sizes (B = 32 dlav1_34, B = 64 dla_34, 512x512, f16x3): one image out of the
    batch against the same image run alone (split-K differs at B = 1: float32 round-off) and against the CPU oracle."""
    heads = synth.HEADS_POSE
    sd = synth.make_state_dict(arch, heads)
    x = synth.frames(B, seed=61)
    model = hip.HipModel(arch, heads, sd, precision="f16x3")
    z = {k: v[pick:pick + 1].cpu() for k, v in model(x.to(device), sigmoid_hm=True).items()}
    z1 = model(x[pick:pick + 1].to(device), sigmoid_hm=True)
    for k in heads:
        assert float((z1[k].cpu() - z[k]).abs().max()) < 2e-4 * max(1.0, float(z[k].abs().max())), k
    zo = ob.dlaseg_forward(sd, x[pick:pick + 1], heads, arch=arch.split("_")[0])
    assert float((z["hm"] - torch.sigmoid(zo["hm"])).abs().max()) < 1e-3
    assert float((z["hm_hp"] - torch.sigmoid(zo["hm_hp"])).abs().max()) < 1e-3
    for k in ("wh", "hps", "reg", "hp_offset", "scale"):
        assert float((z[k] - zo[k]).abs().max()) < 1e-3 * max(1.0, float(zo[k].abs().max())), k


# kernel names (cp_kernel_variant_name) every bench-size forward must have launched: the hot path of SURVEY 8(a) M2-M7
_HOT_VARIANTS = ("halo16_head_f16x3", "halo16_f16x3_m128n128", "halo16_f16x3_m128n64", "halo16_f16x3_m128n32", "pw16_f16x3",
                 "lowc_stem_level0", "lowc_3x3s2", "igemm16_f16x3", "dcn16t_f16x3", "dcn16p_f16x3", "strm16_f16x3")
# (the CenterPoseTrack networks add the previous-frame stems' outputs to the stem's: their first two layers stay two kernels)
# (and at their B = 16 the 128^2 offset convolutions have too few (strip, band) jobs for the row-streaming kernel)
_HOT_VARIANTS_TRACK = tuple(v for v in _HOT_VARIANTS if v not in ("lowc_stem_level0", "strm16_f16x3")) + ("lowc_stem7x7", "lowc_3x3_c16")


@pytest.mark.parametrize("arch,B,reps", [("dla_34", 64, 60), ("dlav1_34", 32, 60)])
def test_backbone_at_bench_batch_every_image_every_launch(device, arch, B, reps):
    """BASELINE configs[2] / configs[1] sizes, f16x3, EVERY image of the batch, `reps` forwards: the class of defect round 4 met in
    dcn16p (wrong values in the last workgroups of a launch larger than the chip, one launch in a hundred) is invisible to a
    one-image, one-launch spot check.  (1) Every forward must be bit-identical to the first; odd forwards run the batch in
    REVERSED image order, so each image is computed by the first workgroups of a launch in one order and by the last in the
    other -- same batch, same |max| pre-scales, same K partition => bit-equal per image; (2) every image must agree with the same
    image inside a batch of 8 (other split-K / tile choices: float32 round-off); (3) the kernels that ran are the hot-path
    ones."""
    heads = synth.HEADS_POSE
    sd = synth.make_state_dict(arch, heads)
    x = synth.frames(B, seed=61).to(device)
    xr = x.flip(0).contiguous()
    model = hip.HipModel(arch, heads, sd, precision="f16x3")
    model.profile(True)
    z0 = {k: v.clone() for k, v in model(x, sigmoid_hm=True).items()}
    torch.cuda.synchronize()
    ran = model.profile_read()
    model.profile(False)
    for want in _HOT_VARIANTS:
        if arch == "dlav1_34" and want == "halo16_head_f16x3":
            continue   # its heads carry GroupNorm between the 3x3 and the 1x1 (GN.py:4-9): the plain 3x3 kernel + gn_final, no fused head
        assert any(name.startswith(want) for name in ran), (want, sorted(ran))
    for k in heads:
        assert torch.isfinite(z0[k]).all(), k
    nbad = {}
    for it in range(1, reps):
        z = model(xr if it & 1 else x, sigmoid_hm=True)
        for k in heads:
            zk = z[k].flip(0) if it & 1 else z[k]
            if not torch.equal(zk, z0[k]):
                d = (zk - z0[k]).abs().amax(dim=(1, 2, 3))
                nbad.setdefault(k, []).append((it, [int(i) for i in d.nonzero().flatten().tolist()], float(d.max())))
    assert not nbad, nbad
    for b0 in range(0, B, 8):
        z8 = model(x[b0:b0 + 8].contiguous(), sigmoid_hm=True)
        for k in heads:
            ref = z0[k][b0:b0 + 8]
            assert float((z8[k] - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max())), (k, b0)


# the three legs of the driver's bench line that are not BASELINE configs[1] / configs[2]: what each must have launched at ITS benchmarked
# size (other kernels than at the 128^2 goldens: dcn16s from 4096 patches, the 128-wide DCN tile, the walk over all twelve heads)
_LEG_VARIANTS = {
    "hourglass": ("halo16_head_f16x3", "halo16_f16x3_m128n128", "igemm16_f16x3", "pw16_f16x3"),
    # (B = 16: 2048 patches of 8 x 16 -- the 64 -> 64 @128^2 layers are over dcn16t's 1024-workgroup threshold, the deeper Cout = 64
    # ones under it (dcn16p), the others on the 128-wide tile; and exactly the count from which a workgroup of the head launch
    # walks every head of its patch)
    "track": _HOT_VARIANTS_TRACK + ("dcn16p_f16x3_p128n64", "dcn16p_f16x3_p128n128"),
    "track_gru": tuple(v for v in _HOT_VARIANTS_TRACK if v != "halo16_head_f16x3") +
                 ("dcn16p_f16x3_p128n64", "dcn16p_f16x3_p128n128", "halo16_gru_f16x3", "gn_final"),
}


@pytest.mark.parametrize("leg,pick", [("hourglass", 5), ("track", 11), ("track_gru", 6)])
def test_timed_legs_at_bench_size_vs_oracle_every_image_every_launch(device, leg, pick):
    """The legs `hourglass` (BASELINE configs[4] as the reference defines it: 2-stack hourglass, 512^2, B = 8), `track` (dla_34,
    CenterPoseTrack inputs, B = 16) and `track_gru` (configs[4] as the reference can run it: dlav1_34 + ConvGRU, all three pre_*
    inputs, B = 16) are timed in the driver's bench line; this is their parity AT THE BENCHMARKED SIZE, on the very pipeline
    object bench.py times (same seeds, same inputs, f16x3):
    (1) one image of the batch against the CPU oracle (large_hourglass.py:266-287, pose_dla_dcn.py:312-318,545-555 restated in
        oracle/) at the north-star gates: 1e-3 on post-sigmoid heat-maps, 1e-3 relative on every regression head, ALL heads;
    (2) 20 forwards, odd ones with the batch (and every pre_* input) in reversed image order: every head of every image
        bit-identical to the first forward (each image is computed by the first workgroups of a launch one way round and by the
        last the other way);
    (3) the launch profile lists the kernels this size really selects."""
    import bench
    from oracle import hourglass as oh

    B = bench.DEFAULT_BATCH[leg]
    pipe = bench.Pipeline(leg, B, device, seed=317, precision="f16x3")
    model, x, extra, heads = pipe.model, pipe.x, pipe.extra, pipe.heads
    assert x.shape == (B, 3, 512, 512) and (set(extra) == {"pre_img", "pre_hm", "pre_hm_hp"}) == (leg != "hourglass")
    model.profile(True)
    z0 = {k: v.clone() for k, v in model(x, sigmoid_hm=True, **extra).items()}
    torch.cuda.synchronize()
    ran = model.profile_read()
    model.profile(False)
    for want in _LEG_VARIANTS[leg]:
        assert any(name.startswith(want) for name in ran), (leg, want, sorted(ran))
    assert set(z0) == set(heads) and len(heads) == (7 if leg == "hourglass" else 11)
    # (1) oracle, one image
    sd = synth.make_state_dict(pipe.arch, heads, pipe.track)
    xi = x[pick:pick + 1].cpu()
    if leg == "hourglass":
        zo = oh.hourglass_forward(sd, xi, heads)
    else:
        zo = ob.dlaseg_forward(sd, xi, heads, arch=pipe.arch.split("_")[0], tracking_task=True,
                               **{k: v[pick:pick + 1].cpu() for k, v in extra.items()})
    for k in heads:
        got = z0[k][pick:pick + 1].cpu()
        assert torch.isfinite(z0[k]).all(), k
        if k in ("hm", "hm_hp"):
            assert float((got - torch.sigmoid(zo[k])).abs().max()) < 1e-3, (leg, k)
        else:
            assert float((got - zo[k]).abs().max()) < 1e-3 * max(1.0, float(zo[k].abs().max())), (leg, k)
    # (2) every image, every launch, both image orders
    xr = x.flip(0).contiguous()
    er = {k: v.flip(0).contiguous() for k, v in extra.items()}
    nbad = {}
    for it in range(1, 20):
        z = model(xr, sigmoid_hm=True, **er) if it & 1 else model(x, sigmoid_hm=True, **extra)
        for k in heads:
            zk = z[k].flip(0) if it & 1 else z[k]
            if not torch.equal(zk, z0[k]):
                d = (zk - z0[k]).abs().amax(dim=(1, 2, 3))
                nbad.setdefault(k, []).append((it, [int(i) for i in d.nonzero().flatten().tolist()], float(d.max())))
    assert not nbad, (leg, nbad)


def _many_launches(f, n):
    first, nbad = None, 0
    for _ in range(n):
        y = f()
        if first is None:
            first = y.clone()
        else:
            nbad += int(not torch.equal(y, first))
    return first, nbad


@pytest.mark.parametrize("name,B,H,Cin,Cout,k,stride,res,n", [
    ("halo16 N=64 (64->64 @128^2 BasicBlock)", 32, 128, 64, 64, 3, 1, True, 200),
    ("halo16 N=128 (256->256 @32^2)", 64, 32, 256, 256, 3, 1, True, 200),
    ("halo16 N=32 (offset convolution 64->27)", 32, 128, 64, 27, 3, 1, False, 200),
    ("strm16 (64->32 @128^2, one band of 16 rows per wave slot)", 64, 128, 64, 32, 3, 1, False, 200),
    ("strm16 (64->32 @128^2, B = 32: bands of 8 rows)", 32, 128, 64, 32, 3, 1, False, 200),
    ("igemm16p stride 2 (64->128 @128^2)", 32, 128, 64, 128, 3, 2, False, 200),
    ("pw16 1x1 (128->128 @64^2)", 64, 64, 128, 128, 1, 1, False, 200),
    ("pw16 1x1 (512->256 @16^2... Root)", 64, 16, 512, 256, 1, 1, True, 200),
    ("exact-f32 igemm, level0's shape (16->16 @512^2; Cin 16 is not a split-f16 shape of the unit-test entry)", 8, 512, 16, 16, 3, 1, False, 100),
    ("exact-f32 igemm, level1's shape (16->32 stride 2)", 8, 512, 16, 32, 3, 2, False, 100),
])
def test_conv_kernels_are_stable_over_many_launches(device, f16x3, name, B, H, Cin, Cout, k, stride, res, n):
    """(The first three layers' lowc kernels have no stand-alone entry: test_first_layers_are_stable_over_many_launches below, and
    60 forwards per architecture in test_backbone_at_bench_batch_every_image_every_launch.)
    Every hot-path convolution kernel family at a launch with more workgroups than the chip holds at once, n launches: each must
    be bit-identical to the first (no launch-to-launch variation), and the first must agree with a float64 convolution of two
    sampled images (the first and the LAST of the batch -- the last workgroups of the launch) to the f16x3 error bound."""
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout)
    x = torch.randn(B, H, H, Cin, generator=g).to(device)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(device)
    affine = Cout % 32 == 0   # (the unit-test entry takes scale / shift only for whole N tiles; the 27-channel offset convolution has a bias only in the engine)
    scale = (0.5 + torch.rand(Cout, generator=g)).to(device) if affine else None
    shift = torch.randn(Cout, generator=g).to(device) if affine else None
    Ho = (H + 2 * (k // 2) - k) // stride + 1
    r = torch.randn(B, Ho, Ho, Cout, generator=g).to(device) if res else None
    f = lambda: hip.conv2d_nhwc(x, w, scale, shift, r, stride, k // 2, 1)
    first, nbad = _many_launches(f, n)
    assert nbad == 0, (name, nbad)
    for b in (0, B - 1):
        ref = F.conv2d(x[b:b + 1].permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), None, stride, k // 2)
        if affine:
            ref = ref * scale.double().cpu().view(1, -1, 1, 1) + shift.double().cpu().view(1, -1, 1, 1)
        if res:
            ref = ref + r[b:b + 1].permute(0, 3, 1, 2).double().cpu()
        ref = ref.clamp_min(0.0)
        got = first[b:b + 1].permute(0, 3, 1, 2).double().cpu()
        assert float((got - ref).abs().max()) < 3e-5 * max(1.0, float(ref.abs().max())), (name, b)


@pytest.mark.parametrize("arch,B,h,w", [("dla_34", 2, 128, 128), ("dla_34", 2, 96, 160), ("dlav1_34", 1, 256, 256), ("dla_34", 3, 512, 512),
                                        ("dla_34", 1, 32, 32), ("dla_34", 5, 64, 224)])
def test_fused_stem_level0_vs_two_kernels_and_oracle(device, arch, B, h, w):
    """lowc2_kernel (stem 7x7 + level0 3x3 in one launch: a wave streams down a 14-column strip, the 16-channel tensor between the
    two layers never exists; its pre-scale comes from a bound, not from a measured |max|) against the two-kernel form
    (cp_set_debug 134217728) on level0's output (tap) -- other pre-scale and another summation order: float32 round-off -- and,
    through the whole network, against the CPU oracle at the north-star gates.  Sizes: strips that end inside the picture (96 x 160:
    12 strips, the last 6 columns wide; 224: exactly 16), one band and several, a picture smaller than a band, B not a power of
    two.  The tap of the stem itself still works (the engine falls back to two kernels for it)."""
    heads = synth.HEADS_POSE
    sd = synth.make_state_dict(arch, heads)
    x = synth.frames(B, seed=23, h=h, w=w).to(device)
    model = hip.HipModel(arch, heads, sd, precision="f16x3")
    model.profile(True)
    z1, t1 = model.forward(x, tap="base.level0")
    ran = model.profile_read()
    model.profile(False)
    assert any(n.startswith("lowc_stem_level0") for n in ran) and not any(n.startswith("lowc_stem7x7") for n in ran), sorted(ran)
    z1 = {k: v.clone() for k, v in z1.items()}
    t1 = t1.clone()
    _, t1b = model.forward(x, tap="base.level0")
    assert torch.equal(t1, t1b)   # deterministic
    hip.lib().cp_set_debug(134217728)
    try:
        model.profile(True)
        _, t0 = model.forward(x, tap="base.level0")
        assert any(n.startswith("lowc_stem7x7") for n in model.profile_read())
        model.profile(False)
    finally:
        hip.lib().cp_set_debug(0)
    assert t1.shape == t0.shape == (B, 16, h, w)
    assert float((t1 - t0).abs().max()) < 2e-6 * max(1.0, float(t0.abs().max()))
    _, ts = model.forward(x, tap="base.base_layer")   # asking for the stem's output selects the two-kernel form
    assert ts.shape == (B, 16, h, w) and float(ts.abs().max()) > 0
    zo = ob.dlaseg_forward(sd, x[B - 1:].cpu(), heads, arch=arch.split("_")[0])
    assert float((torch.sigmoid(z1["hm"][B - 1:]).cpu() - torch.sigmoid(zo["hm"])).abs().max()) < 1e-3
    for k in ("wh", "hps", "reg", "hp_offset", "scale"):
        assert float((z1[k][B - 1:].cpu() - zo[k]).abs().max()) < 1e-3 * max(1.0, float(zo[k].abs().max())), k


@pytest.mark.parametrize("arch,B,h,w", [("dla_34", 2, 128, 128), ("dla_34", 2, 96, 160), ("dlav1_34", 1, 256, 256), ("dla_34", 3, 512, 512),
                                        ("dla_34", 1, 32, 32), ("dla_34", 5, 64, 224), ("dla_34", 2, 160, 352)])
def test_row_streamed_level1_vs_tile_kernel_and_float64(device, arch, B, h, w):
    """lowc1s_kernel (level1, 3x3 / stride 2, 16 -> 32 as a row stream: even / odd input columns in separate LDS planes, two rolling
    accumulators; cp_set_debug 1073741824: at any size, 262144: never) on level1's output (tap) against the tile kernel -- other
    MFMA shape, other summation order: float32 round-off -- and against a float64 convolution of level0's output with the folded
    BatchNorm + ReLU (pose_dla_dcn.py:310-322).  Sizes: strips that end inside the picture (80 / 112 / 176 output columns), one band
    and several, pictures smaller than a strip, several jobs per wave."""
    heads = synth.HEADS_POSE
    sd = synth.make_state_dict(arch, heads)
    x = synth.frames(B, seed=29, h=h, w=w).to(device)
    model = hip.HipModel(arch, heads, sd, precision="f16x3")
    outs = {}
    for name, dbg, want in (("rows", 1073741824, "lowc_3x3s2_c16_rows"), ("tile", 262144, "lowc_3x3s2_c16_f16x3")):
        hip.lib().cp_set_debug(dbg)
        try:
            model.profile(True)
            _, t = model.forward(x, tap="base.level1")
            ran = model.profile_read()
            model.profile(False)
            outs[name] = t.clone()
            _, t2 = model.forward(x, tap="base.level1")
            assert torch.equal(outs[name], t2), name   # deterministic
        finally:
            hip.lib().cp_set_debug(0)
        assert any(n.startswith(want) for n in ran), (name, sorted(ran))
    _, l0 = model.forward(x, tap="base.level0")
    wgt = sd["base.level1.0.weight"].double()
    bn = {k: sd["base.level1.1." + k].double() for k in ("weight", "bias", "running_mean", "running_var")}
    ref = F.conv2d(l0.double().cpu(), wgt, None, 2, 1)
    scale = bn["weight"] / torch.sqrt(bn["running_var"] + 1e-5)
    ref = F.relu(ref * scale.view(1, -1, 1, 1) + (bn["bias"] - bn["running_mean"] * scale).view(1, -1, 1, 1))
    assert outs["rows"].shape == outs["tile"].shape == ref.shape == (B, 32, (h - 1) // 2 + 1, (w - 1) // 2 + 1)
    top = max(1.0, float(ref.abs().max()))
    assert float((outs["rows"].double().cpu() - ref).abs().max()) < 2e-5 * top
    assert float((outs["rows"] - outs["tile"]).abs().max()) < 4e-6 * top


def test_first_layers_are_stable_over_many_launches(device):
    """lowc.hip (7x7 stem, level0, level1 -- 8 % of the step) at launches larger than the chip: the level1 activation of a batch of
    16 frames (tap), 100 forwards, odd ones with the batch in reversed image order: per image bit-identical to the first forward."""
    heads = synth.HEADS_POSE
    model = hip.HipModel("dla_34", heads, synth.make_state_dict("dla_34", heads), precision="f16x3")
    x = synth.frames(16, seed=5).to(device)
    xr = x.flip(0).contiguous()
    _, first = model.forward(x, tap="base.level1")
    first = first.clone()
    assert first.shape == (16, 32, 256, 256) and float(first.abs().max()) > 0
    for it in range(100):
        _, t = model.forward(xr if it % 2 else x, tap="base.level1")
        t = t.flip(0) if it % 2 else t
        assert torch.equal(t, first), it


@pytest.mark.parametrize("arch,B", [("dla_34", 1), ("dlav1_34", 2), ("hourglass", 1)])
def test_splitk_epilogue_quad_form_equals_elementwise(device, arch, B):
    """Small launches are cut along K by the engine (batch 1 - 2: 48 of a dla_34 frame's 138 launches are split-K epilogues);
    the slices' slabs are summed by splitk_epilogue.  Its quad form (four channels per lane, every slab read in flight at once)
    against the element-wise form (cp_set_debug 8) on whole networks: the same additions in slice order -> every head
    bit-identical, 10 forwards each way.  (cp_conv2d_nhwc never splits K, so this cannot be a single-layer test.)"""
    heads = synth.HEADS_POSE
    model = hip.HipModel(arch, heads, synth.make_state_dict(arch, heads), precision="f16x3")
    x = synth.frames(B, seed=11).to(device)
    first = {k: v.clone() for k, v in model(x, sigmoid_hm=True).items()}
    for it in range(20):
        hip.lib().cp_set_debug(8 if it % 2 == 0 else 0)
        try:
            z = model(x, sigmoid_hm=True)
        finally:
            hip.lib().cp_set_debug(0)
        for name in first:
            assert torch.equal(z[name], first[name]), (it, name)


@pytest.mark.parametrize("arch,B,hw,tracking", [("dla_34", 2, 256, False), ("hourglass", 1, 512, False), ("dla_34", 1, 512, False),
                                                ("dla_34", 16, 512, False), ("dla_34", 16, 512, True)])
def test_grouped_fused_heads_equal_per_head_launches(device, arch, B, hw, tracking):
    """All fused prediction heads in ONE launch (engine.hip: fused_heads_grouped; concatenated operands, per-tile head
    table; a workgroup walks the hidden tiles of its head and finishes the maps) against one launch per head and against the
    slab + reduction-launch form: the same arithmetic in the same order -> bit-identical head tensors, sigmoid included."""
    # (tracking: the twelve heads of the CenterPoseTrack network -- the most a grouped launch walks -- on the two-frame inputs)
    heads = synth.HEADS_TRACK if tracking else synth.HEADS_POSE
    sd = synth.make_state_dict(arch, heads, tracking)
    x, kw = mg.backbone_inputs(tracking, res=hw, seed=77, batch=B)
    x, kw = x.to(device), {k: v.to(device) for k, v in kw.items()}
    model = hip.HipModel(arch, heads, sd, tracking_task=tracking, precision="f16x3")
    run = lambda: {k: v.clone() for k, v in model(x, sigmoid_hm=True, **kw).items()}
    z = run()
    # 16777216: one launch per head; 1: one grouped launch that writes per-tile slabs + the reduction launch; 2: a workgroup
    # walks the hidden tiles of ONE head and finishes its maps (the default below 2048 patches; from there -- the B = 16 case
    # here -- a workgroup walks every head of its patch)
    for dbg in (16777216, 1, 2):
        hip.lib().cp_set_debug(dbg)
        try:
            z1 = run()
        finally:
            hip.lib().cp_set_debug(0)
        for k in heads:
            assert torch.isfinite(z[k]).all(), k
            assert torch.equal(z[k], z1[k]), (dbg, k, float((z[k] - z1[k]).abs().max()))


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("arch,tracking", CONFIGS)
def test_backbone_vs_reference_golden(device, arch, tracking, prec):
    """Head tensors against the REFERENCE modules' outputs (tests/golden/backbone_*.npz), both arithmetic modes."""
    heads = synth.HEADS_TRACK if tracking else synth.HEADS_POSE
    gold = np.load(os.path.join(GOLD, "backbone_%s.npz" % synth.config_key(arch, tracking)))
    sd = synth.make_state_dict(arch, heads, tracking)
    x, kw = mg.backbone_inputs(tracking)
    model = hip.HipModel(arch, heads, sd, tracking_task=tracking, precision=prec)
    z = model(x.to(device), **{k: v.to(device) for k, v in kw.items()})
    for k in heads:
        ref = gold[k]
        np.testing.assert_allclose(z[k].cpu().numpy(), ref, rtol=0, atol=1e-3 * max(1.0, np.abs(ref).max()), err_msg=k)
    for k in ("hm", "hm_hp"):  # north-star gate: <= 1e-3 on the (post-sigmoid) heat-maps
        np.testing.assert_allclose(torch.sigmoid(z[k]).cpu().numpy(), 1 / (1 + np.exp(-gold[k])), rtol=0, atol=1e-3)


@pytest.mark.parametrize("arch,tracking", CONFIGS)
def test_backbone_with_row_streamed_offset_convolutions_vs_reference_golden(device, arch, tracking):
    """The reference goldens again with the 64-channel conv_offset_mask layers forced onto strm16.hip at the goldens' small size
    (cp_set_debug 536870912; by default the kernel only takes them from B = 32 at 128 x 128): 27 of 32 channels, the mask sigmoid
    from channel 18 (dcn_v2.py:105-125), few jobs per workgroup.  Same gates as test_backbone_vs_reference_golden, and against the
    default dispatch the heads may differ only by what the sigmoid form (exp2 / rcp, < 3e-7 per mask) propagates."""
    heads = synth.HEADS_TRACK if tracking else synth.HEADS_POSE
    gold = np.load(os.path.join(GOLD, "backbone_%s.npz" % synth.config_key(arch, tracking)))
    sd = synth.make_state_dict(arch, heads, tracking)
    x, kw = mg.backbone_inputs(tracking)
    model = hip.HipModel(arch, heads, sd, tracking_task=tracking, precision="f16x3")
    args = (x.to(device),), {k: v.to(device) for k, v in kw.items()}
    z0 = {k: v.clone() for k, v in model(*args[0], **args[1]).items()}
    hip.lib().cp_set_debug(536870912)
    try:
        model.profile(True)
        z = {k: v.clone() for k, v in model(*args[0], **args[1]).items()}
        torch.cuda.synchronize()
        ran = model.profile_read()
        model.profile(False)
    finally:
        hip.lib().cp_set_debug(0)
    assert any(name.startswith("strm16_f16x3") for name in ran), sorted(ran)
    for k in heads:
        ref = gold[k]
        np.testing.assert_allclose(z[k].cpu().numpy(), ref, rtol=0, atol=1e-3 * max(1.0, np.abs(ref).max()), err_msg=k)
        assert float((z[k] - z0[k]).abs().max()) < 2e-5 * max(1.0, float(z0[k].abs().max())), k
    for k in ("hm", "hm_hp"):
        np.testing.assert_allclose(torch.sigmoid(z[k]).cpu().numpy(), 1 / (1 + np.exp(-gold[k])), rtol=0, atol=1e-3)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("arch", ["dla_34", "dlav1_34"])
def test_backbone_512_vs_oracle_and_batch_invariance(device, arch, prec):
    heads = synth.HEADS_POSE
    sd = synth.make_state_dict(arch, heads)
    x = synth.frames(3, seed=23)
    model = hip.HipModel(arch, heads, sd, precision=prec)
    z = model(x.to(device), sigmoid_hm=True)
    zo = ob.dlaseg_forward(sd, x[1:2], heads, arch=arch.split("_")[0])
    assert float((z["hm"][1:2].cpu() - torch.sigmoid(zo["hm"])).abs().max()) < 1e-3
    assert float((z["hm_hp"][1:2].cpu() - torch.sigmoid(zo["hm_hp"])).abs().max()) < 1e-3
    for k in ("wh", "hps", "reg", "hp_offset", "scale"):
        assert float((z[k][1:2].cpu() - zo[k]).abs().max()) < 1e-3 * max(1.0, float(zo[k].abs().max())), k
    # size-independent property: images are independent -> the batch is a pure stack.  Bit-exact when both runs
    # use the same K partition; at batch 1 the low-resolution layers switch to split-K (different summation
    # order), so the comparison across batch sizes is to float32 round-off.
    z1 = model(x[1:2].to(device), sigmoid_hm=True)
    for k in heads:
        assert float((z1[k] - z[k][1:2]).abs().max()) < 2e-4 * max(1.0, float(z[k].abs().max())), k
    z3 = model(x.to(device), sigmoid_hm=True)
    for k in heads:
        assert torch.equal(z3[k], z[k]), k   # same shape twice: deterministic, bit-exact


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_backbone_non_square_ragged_tiles_vs_oracle(device, prec):
    """96 x 160 input: the last tiles of lowc.hip (64/32-pixel wide), of the implicit GEMMs (M % 128 != 0 at the
    deeper levels) and the split-K path are all ragged here."""
    heads = synth.HEADS_POSE
    sd = synth.make_state_dict("dla_34", heads)
    x = synth.frames(2, seed=31, h=96, w=160)
    model = hip.HipModel("dla_34", heads, sd, precision=prec)
    z = model(x.to(device), sigmoid_hm=True)
    zo = ob.dlaseg_forward(sd, x, heads, arch="dla")
    assert z["hm"].shape == (2, 1, 24, 40)
    assert float((z["hm"].cpu() - torch.sigmoid(zo["hm"])).abs().max()) < 1e-3
    for k in ("wh", "hps", "reg", "hp_offset", "scale"):
        assert float((z[k].cpu() - zo[k]).abs().max()) < 1e-3 * max(1.0, float(zo[k].abs().max())), k


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_hourglass_vs_reference_golden(device, prec):
    """Stacked hourglass (BASELINE configs[4] as the reference defines it, single frame): engine vs the reference
    module's own output on the seeded weights (tests/golden/backbone_hourglass.npz, 128x128 -> 1x1 at the deepest level)."""
    heads = synth.HEADS_POSE
    gold = np.load(os.path.join(GOLD, "backbone_hourglass.npz"))
    sd = synth.make_state_dict("hourglass", heads)
    x, _ = mg.backbone_inputs(False)
    model = hip.HipModel("hourglass", heads, sd, precision=prec)
    z = model(x.to(device))
    for k in heads:
        ref = torch.from_numpy(gold[k])
        err = float((z[k].cpu() - ref).abs().max())
        assert err < 1e-3 * max(1.0, float(ref.abs().max())), (k, err)
    hm = torch.sigmoid(z["hm"].cpu())
    assert float((hm - torch.sigmoid(torch.from_numpy(gold["hm"]))).abs().max()) < 1e-3


def test_hourglass_256_vs_oracle_and_rejects_bad_sizes(device):
    from oracle import hourglass as oh

    heads = synth.HEADS_POSE
    sd = synth.make_state_dict("hourglass", heads)
    x = synth.frames(2, seed=37, h=256, w=256)
    model = hip.HipModel("hourglass", heads, sd, precision="f16x3")
    z = model(x.to(device), sigmoid_hm=True)
    zo = oh.hourglass_forward(sd, x, heads)
    assert float((z["hm"].cpu() - torch.sigmoid(zo["hm"])).abs().max()) < 1e-3
    for k in ("wh", "hps", "reg", "hp_offset", "scale"):
        assert float((z[k].cpu() - zo[k]).abs().max()) < 1e-3 * max(1.0, float(zo[k].abs().max())), k
    with pytest.raises(RuntimeError):
        model(synth.frames(1, seed=1, h=160, w=160).to(device))   # not a multiple of 128


def test_fused_head_matches_unfused_path(device):
    """dla_34 heads (conv3x3 -> ReLU -> conv1x1, no GroupNorm) run as one fused kernel in f16x3 mode; the two-kernel
    path (debug flag 32) must give the same maps to float32 round-off, and the fused path must be deterministic."""
    heads = synth.HEADS_POSE
    sd = synth.make_state_dict("dla_34", heads)
    x = synth.frames(2, seed=29, h=256, w=256).to(device)
    model = hip.HipModel("dla_34", heads, sd, precision="f16x3")
    model.profile(True)
    model(x, sigmoid_hm=True)
    prof = model.profile_read()
    # the fused kernel is what runs: on the halo-resident kernel for maps that tile into 8x16 patches (64x64 here)
    assert "halo16_head_f16x3_m128n128" in prof or "igemm16_head_f16x3_m128n128" in prof, list(prof)
    model.profile(False)
    z = {k: v.clone() for k, v in model(x, sigmoid_hm=True).items()}
    z2 = model(x, sigmoid_hm=True)
    hip.lib().cp_set_debug(32)
    try:
        zu = {k: v.clone() for k, v in model(x, sigmoid_hm=True).items()}
    finally:
        hip.lib().cp_set_debug(0)
    hip.lib().cp_set_debug(4096)   # the same fusion on the per-tap implicit-GEMM kernel (what ragged maps fall back to)
    try:
        model.profile(True)
        zi = {k: v.clone() for k, v in model(x, sigmoid_hm=True).items()}
        assert "igemm16_head_f16x3_m128n128" in model.profile_read()
        model.profile(False)
    finally:
        hip.lib().cp_set_debug(0)
    for k in heads:
        assert torch.equal(z[k], z2[k]), k
        assert z[k].shape == zu[k].shape
        assert float((z[k] - zu[k]).abs().max()) < 2e-5 * max(1.0, float(zu[k].abs().max())), k
        assert float((z[k] - zi[k]).abs().max()) < 2e-5 * max(1.0, float(zu[k].abs().max())), k


def test_convgru_step_kernels_agree(device):
    """dlav1_34's ConvGRU hidden-side step (convGRU.py:32-39) has three implementations: the gate arithmetic fused into
    the halo-resident kernel (maps that tile into 8x16 patches), into the per-tap implicit-GEMM kernel (cp_set_debug 4096,
    what ragged maps fall back to) and the unfused convolution + gate kernel (256).  Same products, different summation
    orders: the heads must agree to float32 round-off."""
    heads = synth.HEADS_POSE
    sd = synth.make_state_dict("dlav1_34", heads)
    x = synth.frames(2, seed=31, h=256, w=256).to(device)
    model = hip.HipModel("dlav1_34", heads, sd, precision="f16x3")
    model.profile(True)
    z = {k: v.clone() for k, v in model(x, sigmoid_hm=True).items()}
    assert "halo16_gru_f16x3_m128n96" in model.profile_read()
    outs = {}
    for flag, name in ((4096, "igemm16_gru_f16x3_m128n96"), (256, None)):
        hip.lib().cp_set_debug(flag)
        try:
            outs[flag] = {k: v.clone() for k, v in model(x, sigmoid_hm=True).items()}
            if name:
                assert name in model.profile_read()
        finally:
            hip.lib().cp_set_debug(0)
    model.profile(False)
    for k in heads:
        for flag in outs:
            tol = 2e-5 * max(1.0, float(z[k].abs().max()))
            assert float((z[k] - outs[flag][k]).abs().max()) < tol, (k, flag)


def test_previous_frame_stems_are_independent(device):
    """Each pre_* stem exists iff its own flag was set (pose_dla_dcn.py:253-271): a checkpoint with only pre_img_layer
    loads, uses pre_img, and refuses a pre_hm it has no layer for (instead of ignoring it)."""
    heads = synth.HEADS_TRACK
    sd = synth.make_state_dict("dla_34", heads, True)
    sd = {k: v for k, v in sd.items() if "pre_hm_layer" not in k and "pre_hm_hp_layer" not in k}
    x, kw = mg.backbone_inputs(True)
    model = hip.HipModel("dla_34", heads, sd, tracking_task=True, precision="f16x3")
    z = model(x.to(device), pre_img=kw["pre_img"].to(device), sigmoid_hm=True)
    zo = ob.dlaseg_forward(sd, x, heads, arch="dla", tracking_task=True, pre_img=kw["pre_img"])
    assert float((z["hm"].cpu() - torch.sigmoid(zo["hm"])).abs().max()) < 1e-3
    assert float((z["tracking"].cpu() - zo["tracking"]).abs().max()) < 1e-3 * max(1.0, float(zo["tracking"].abs().max()))
    with pytest.raises(RuntimeError, match="pre_"):
        model(x.to(device), pre_img=kw["pre_img"].to(device), pre_hm=kw["pre_hm"].to(device))


def test_model_missing_parameter_fails_loudly(device):
    sd = synth.make_state_dict("dla_34")
    del sd["base.level3.tree1.root.conv.weight"]
    with pytest.raises(RuntimeError, match="root.conv.weight"):
        hip.HipModel("dla_34", synth.HEADS_POSE, sd)


def _decode_gpu(d, device, tracking, sem, rep_mode=1, K=100, apply_sigmoid=False):
    g = {k: torch.from_numpy(v).to(device) for k, v in d.items()}
    det = hip.decode_raw(g["hm"], g["hps"], g["wh"], g["hm_hp"], g.get("hps_uncertainty"), g["scale"],
                         g.get("scale_uncertainty"), g["reg"], g["hp_offset"], g.get("tracking"), g.get("tracking_hp"),
                         K=K, rep_mode=rep_mode, fit_gaussian=tracking, balance=2.0,
                         legacy_bool_mask=(sem == "bool"), apply_sigmoid=apply_sigmoid)
    return {k: v.numpy() for k, v in hip.split_detections(det.cpu()).items()}, g


@pytest.mark.parametrize("name,tracking,sem,B,seed,rep", [
    ("decode_pose_bool", False, "bool", 2, 317, 1),
    ("decode_pose_uint8", False, "uint8", 2, 317, 1),
    ("decode_track_uint8", True, "uint8", 1, 317, 1),
    ("decode_pose_uint8_rep0", False, "uint8", 1, 318, 0),
])
def test_decode_vs_reference_golden(device, name, tracking, sem, B, seed, rep):
    """Device decode against the outputs of the REFERENCE's object_pose_decode (both mask semantics)."""
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    d = odec.synth_heads(B, seed=seed, tracking=tracking)
    r, _ = _decode_gpu(d, device, tracking, sem, rep)
    for k in gold.files:
        if k in ("kps_displacement_std", "obj_scale_uncertainty"):
            np.testing.assert_allclose(r[k], gold[k], rtol=3e-6, atol=0, err_msg=k)
        elif tracking and k.startswith("kps_heatmap"):
            np.testing.assert_allclose(r[k], gold[k], rtol=1e-6, atol=1e-6, err_msg=k)
        else:
            np.testing.assert_array_equal(r[k], gold[k], err_msg=k)


def test_decode_batch32_full_size_vs_oracle_and_edge_cases(device):
    # BASELINE config 2 size: B = 32 at 128x128
    d = odec.synth_heads(32, seed=99)
    r, _ = _decode_gpu(d, device, False, "uint8")
    o = odec.object_pose_decode(d["hm"], d["hps"], wh=d["wh"], obj_scale=d["scale"], reg=d["reg"], hm_hp=d["hm_hp"],
                                hp_offset=d["hp_offset"], K=100, rep_mode=1)
    for k in o:
        np.testing.assert_array_equal(r[k], o[k], err_msg=k)
    # properties that hold at any size: scores sorted, indices in range, invalid entries are exactly -10000
    s = r["scores"][..., 0]
    assert (np.diff(s, axis=1) <= 0).all()
    hm = r["kps_heatmap_mean"]
    assert ((hm == -10000) | ((hm > -1) & (hm < 129))).all()
    # sparse maps: fewer than K peaks -> the tail is zeros in index order (value desc, index asc)
    d2 = odec.synth_heads(1, seed=5)
    d2["hm"] = d2["hm"] * (d2["hm"] > 0.9)
    d2["hm_hp"] = d2["hm_hp"] * (d2["hm_hp"] > 0.9)
    r2, _ = _decode_gpu(d2, device, False, "uint8")
    o2 = odec.object_pose_decode(d2["hm"], d2["hps"], wh=d2["wh"], obj_scale=d2["scale"], reg=d2["reg"],
                                 hm_hp=d2["hm_hp"], hp_offset=d2["hp_offset"], K=100, rep_mode=1)
    for k in o2:
        np.testing.assert_array_equal(r2[k], o2[k], err_msg="sparse " + k)
    # optional heads absent (reg / hp_offset None -> +0.5 rule, zero-filled records)
    g = {k: torch.from_numpy(v).to(device) for k, v in odec.synth_heads(1, seed=6).items()}
    det = hip.decode_raw(g["hm"], g["hps"], g["wh"], g["hm_hp"], K=40, rep_mode=1)
    dd = odec.synth_heads(1, seed=6)
    o3 = odec.object_pose_decode(dd["hm"], dd["hps"], wh=dd["wh"], hm_hp=dd["hm_hp"], K=40, rep_mode=1)
    r3 = {k: v.numpy() for k, v in hip.split_detections(det.cpu()).items()}
    for k in o3:
        np.testing.assert_array_equal(r3[k], o3[k], err_msg="no-optional " + k)


def test_decode_fused_sigmoid_and_rejects_bad_shapes(device):
    d = odec.synth_heads(1, seed=8)
    logit = lambda p: np.log(np.clip(p, 1e-6, 1 - 1e-6) / (1 - np.clip(p, 1e-6, 1 - 1e-6))).astype(np.float32)
    d_l = dict(d, hm=logit(d["hm"]), hm_hp=logit(d["hm_hp"]))
    r, g = _decode_gpu(d_l, device, False, "uint8", apply_sigmoid=True)
    # the maps were overwritten with their sigmoid (object_pose.py:136-138) and decode used those values
    hm_s = g["hm"].cpu().numpy()
    assert np.abs(hm_s - 1 / (1 + np.exp(-d_l["hm"]))).max() < 1e-6
    o = odec.object_pose_decode(hm_s, d["hps"], wh=d["wh"], obj_scale=d["scale"], reg=d["reg"],
                                hm_hp=g["hm_hp"].cpu().numpy(), hp_offset=d["hp_offset"], K=100, rep_mode=1)
    for k in o:
        np.testing.assert_array_equal(r[k], o[k], err_msg=k)
    big = torch.zeros(1, 1, 256, 256, device=device)
    with pytest.raises(RuntimeError):
        hip.decode_raw(big, torch.zeros(1, 16, 256, 256, device=device), torch.zeros(1, 2, 256, 256, device=device),
                       torch.zeros(1, 8, 256, 256, device=device))


def _pnp_cases(N, noise, seed, npts=16, drop=0.0):
    rng = np.random.RandomState(seed)
    K = np.array([[663.0287679036459, 0, 300.2775065104167], [0, 663.0287679036459, 395.00066121419275], [0, 0, 1]])
    pts = np.zeros((N, npts, 2), np.float32)
    scale = np.zeros((N, 3), np.float32)
    poses = []
    for i in range(N):
        sc = np.array([rng.uniform(0.3, 3), rng.uniform(0.5, 2.0), rng.uniform(0.3, 3)])
        V = opnp.cuboid_vertices(sc / sc[1])
        q = rng.randn(4)
        R = opnp.quat_xyzw_to_matrix(q / np.linalg.norm(q))
        t = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(6.0, 12.0)])
        uv = opnp.project_points(V, opnp.matrix_to_rodrigues(R), t, K)
        p = np.repeat(uv, npts // 8, axis=0) + rng.randn(npts, 2) * noise
        if drop > 0:
            dead = rng.rand(npts) < drop
            dead[0::2] = False
            p[dead] = -10000
        pts[i], scale[i] = p, sc
        poses.append((R, t))
    return K, pts, scale, poses


def _geodesic_deg(Ra, Rb):
    return np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


@pytest.mark.parametrize("noise,npts,drop", [(0.0, 16, 0.0), (1.0, 16, 0.0), (1.0, 16, 0.3), (0.5, 8, 0.0)])
def test_pnp_known_pose_and_vs_float64_oracle(device, noise, npts, drop):
    N = 96
    K, pts, scale, poses = _pnp_cases(N, noise, seed=int(noise * 10) + npts, npts=npts, drop=drop)
    cam = np.tile(np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]), (N, 1))
    out = hip.pnp_solve(torch.from_numpy(pts).to(device), torch.from_numpy(scale).to(device),
                        torch.from_numpy(cam).to(device)).cpu().numpy()
    assert (out[:, 0] >= 1).all()
    for i in range(N):
        s = opnp.solve_cuboid_pnp(pts[i].astype(np.float64), scale[i].astype(np.float64), K, opencv_return=True)
        if s is None:  # converged behind the camera (tiny far object + noise): both sides must reject it
            assert out[i, 0] == 2
            continue
        assert out[i, 0] == 1
        Rg = opnp.rodrigues_to_matrix(out[i, 1:4])
        # (4) of SURVEY 8(d): <= 1 deg / 1 % vs the float64 restatement (observed ~1e-6 deg / 1e-10)
        assert _geodesic_deg(Rg, opnp.rodrigues_to_matrix(s["rvec"])) < 1e-3
        assert np.linalg.norm(out[i, 4:7] - s["tvec"]) / np.linalg.norm(s["tvec"]) < 1e-6
        np.testing.assert_allclose(out[i, 8:24].reshape(8, 2), s["projected_points"], atol=1e-4)
        if noise == 0.0:  # by construction: the generating pose is recovered
            R, t = poses[i]
            assert _geodesic_deg(Rg, R) < 1e-3
            assert np.linalg.norm(out[i, 4:7] - t) / np.linalg.norm(t) < 1e-6
        # OpenGL-frame outputs (cuboid_pnp_solver.py:179-196)
        s2 = opnp.solve_cuboid_pnp(pts[i].astype(np.float64), scale[i].astype(np.float64), K, opencv_return=False)
        np.testing.assert_allclose(out[i, 28:31], s2["location"], rtol=1e-6, atol=1e-9)
        q = out[i, 31:35] * np.sign(np.dot(out[i, 31:35], s2["quaternion_xyzw"]))
        np.testing.assert_allclose(q, s2["quaternion_xyzw"], atol=1e-6)


def _dlt_gap(pts, scale, K):
    """lambda_1 / lambda_2 of the DLT normal matrix cv's non-planar initialisation takes its smallest eigenvector of
    (oracle/pnp.py: dlt_init) -- the convergence ratio of the device's inverse iteration."""
    verts = opnp.cuboid_vertices(np.asarray(scale, np.float64) / scale[1])
    obj = np.repeat(verts, len(pts) // 8, axis=0)
    mn = np.stack([(pts[:, 0] - K[0, 2]) / K[0, 0], (pts[:, 1] - K[1, 2]) / K[1, 1]], 1).astype(np.float64)
    L = np.zeros((2 * len(pts), 12))
    for i in range(len(pts)):
        X, Y, Z = obj[i]
        x, y = -mn[i, 0], -mn[i, 1]
        L[2 * i] = [X, Y, Z, 1, 0, 0, 0, 0, x * X, x * Y, x * Z, x]
        L[2 * i + 1] = [0, 0, 0, 0, X, Y, Z, 1, y * X, y * Y, y * Z, y]
    w = np.linalg.eigvalsh(L.T @ L)
    return w[0] / w[1]


@pytest.mark.gpu
def test_pnp_slowly_converging_dlt_vs_float64_oracle(device):
    """Point sets whose two smallest DLT eigenvalues are close (ratio > 0.75: plain inverse iteration would need more than
    the 48 steps after which pnp.hip finishes with a Rayleigh-Ritz step over its last two iterates): the pose must still be
    the float64 restatement's, which takes the exact eigenvector (numpy eigh)."""
    # far objects under 2 px of noise: ~4 % of these have a gap ratio above 0.75
    rng = np.random.RandomState(4242)
    K = np.array([[663.0287679036459, 0, 300.2775065104167], [0, 663.0287679036459, 395.00066121419275], [0, 0, 1]])
    M = 1500
    pts = np.zeros((M, 16, 2), np.float32)
    scale = np.zeros((M, 3), np.float32)
    for i in range(M):
        sc = np.array([rng.uniform(0.3, 3), rng.uniform(0.5, 2.0), rng.uniform(0.3, 3)])
        q = rng.randn(4)
        R = opnp.quat_xyzw_to_matrix(q / np.linalg.norm(q))
        t = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(20.0, 60.0)])
        uv = opnp.project_points(opnp.cuboid_vertices(sc / sc[1]), opnp.matrix_to_rodrigues(R), t, K)
        pts[i], scale[i] = np.repeat(uv, 2, axis=0) + rng.randn(16, 2) * 2.0, sc
    gaps = np.array([_dlt_gap(pts[i].astype(np.float64), scale[i], K) for i in range(M)])
    sel = np.argsort(-gaps)[:48]
    assert gaps[sel].min() > 0.75, "the generator no longer produces slowly converging cases (%.3f)" % gaps[sel].min()
    pts, scale = pts[sel], scale[sel]
    N = len(sel)
    cam = np.tile(np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]), (N, 1))
    out = hip.pnp_solve(torch.from_numpy(pts).to(device), torch.from_numpy(scale).to(device),
                        torch.from_numpy(cam).to(device)).cpu().numpy()
    n_pose = 0
    for i in range(N):
        s = opnp.solve_cuboid_pnp(pts[i].astype(np.float64), scale[i].astype(np.float64), K, opencv_return=True)
        if s is None:
            assert out[i, 0] == 2
            continue
        assert out[i, 0] == 1
        assert _geodesic_deg(opnp.rodrigues_to_matrix(out[i, 1:4]), opnp.rodrigues_to_matrix(s["rvec"])) < 1e-3
        assert np.linalg.norm(out[i, 4:7] - s["tvec"]) / np.linalg.norm(s["tvec"]) < 1e-6
        np.testing.assert_allclose(out[i, 8:24].reshape(8, 2), s["projected_points"], atol=1e-4)
        n_pose += 1
    assert n_pose >= 8, "too few of the selected cases have a pose in front of the camera (%d)" % n_pose  # measured: 16 of 48


def test_pnp_status_codes_and_rare_branches(device):
    """< 4 points -> -1; 5 valid points -> EPnP (cuboid_pnp_solver.py:162-163); one cuboid face -> the planar
    (homography) initialisation; 4 points on two vertices -> degenerate, failure (0).  Poses vs the float64 oracle."""
    K, pts, scale, poses = _pnp_cases(6, 0.0, seed=3)
    pts[0, :] = -10000                                   # no valid point            -> -1
    pts[1, 4:] = -10000                                  # 4 points, 2 distinct vertices: degenerate -> 0
    pts[2, 8:] = -10000                                  # one face (vertices 0-3), each twice: planar
    keep = [0, 2, 5, 9, 14]                              # 5 points on 5 vertices -> EPnP
    pts[3, [k for k in range(16) if k not in keep]] = -10000
    pts[4, 8:] = -10000
    pts[4, 0:8:2] = -10000                               # one face, each vertex once: 4 coplanar points go to EPnP
                                                         # (< 6), which has no barycentric coordinates for them -> 0
    cam = np.tile(np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]), (6, 1))
    out = hip.pnp_solve(torch.from_numpy(pts).to(device), torch.from_numpy(scale).to(device),
                        torch.from_numpy(cam).to(device)).cpu().numpy()
    assert list(out[:, 0]) == [-1, 0, 1, 1, 0, 1]
    assert list(out[:, 35]) == [0, 4, 8, 5, 4, 16]
    assert opnp.solve_cuboid_pnp(pts[4].astype(np.float64), scale[4].astype(np.float64), K) is None   # oracle agrees
    for i in (2, 3, 5):
        R, t = poses[i]
        assert _geodesic_deg(opnp.rodrigues_to_matrix(out[i, 1:4]), R) < 1e-3, i      # exact data: the generating pose
        assert np.linalg.norm(out[i, 4:7] - t) / np.linalg.norm(t) < 1e-6, i
        s = opnp.solve_cuboid_pnp(pts[i].astype(np.float64), scale[i].astype(np.float64), K, opencv_return=True)
        assert _geodesic_deg(opnp.rodrigues_to_matrix(out[i, 1:4]), opnp.rodrigues_to_matrix(s["rvec"])) < 1e-3
        np.testing.assert_allclose(out[i, 8:24].reshape(8, 2), s["projected_points"], atol=1e-3)


@pytest.mark.parametrize("kind", ["planar", "epnp5"])
def test_pnp_rare_branches_noisy_vs_oracle(device, kind):
    """1 px noise.  Planar: the same LM optimum as the oracle (<= 1e-3 deg).  EPnP with 5 points is un-refined and picks
    the best of three linearisations, so under noise two implementations of the same published algorithm (normal
    equations + Jacobi on the device, SVD least squares + LAPACK in numpy) may legitimately settle on different
    candidates; what is compared is what the reference consumes next -- the reprojection of the given points -- plus
    the pose whenever both picked the same candidate."""
    N = 48
    K, pts, scale, _ = _pnp_cases(N, 1.0, seed=5)
    if kind == "planar":
        pts[:, 8:] = -10000
    else:
        keep = [0, 2, 5, 9, 14]
        pts[:, [k for k in range(16) if k not in keep]] = -10000
    cam = np.tile(np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]), (N, 1))
    out = hip.pnp_solve(torch.from_numpy(pts).to(device), torch.from_numpy(scale).to(device),
                        torch.from_numpy(cam).to(device)).cpu().numpy()
    n_cmp = n_same = 0
    for i in range(N):
        s = opnp.solve_cuboid_pnp(pts[i].astype(np.float64), scale[i].astype(np.float64), K, opencv_return=True)
        if s is None:
            assert out[i, 0] in (0, 2)
            continue
        assert out[i, 0] in (1, 2), i
        dR = _geodesic_deg(opnp.rodrigues_to_matrix(out[i, 1:4]), opnp.rodrigues_to_matrix(s["rvec"]))
        dt = np.linalg.norm(out[i, 4:7] - s["tvec"]) / np.linalg.norm(s["tvec"])
        if kind == "planar":
            assert out[i, 0] == 1 and dR < 1e-3 and dt < 1e-6, (i, dR, dt)
        else:
            assert out[i, 7] <= max(2.0 * s["reproj_err"], 3.0), (i, out[i, 7], s["reproj_err"])
            n_same += int(dR < 1.0 and dt < 1e-2)
        n_cmp += 1
    if kind != "planar":
        assert n_same >= n_cmp // 2, (n_same, n_cmp)
    assert n_cmp > N // 2


def test_detect_one_call_and_graph_replay(device):
    """cp_model_detect == cp_model_forward + cp_decode, eagerly and replayed from a hipGraph on fresh frames."""
    heads = synth.HEADS_POSE
    sd = synth.make_state_dict("dlav1_34", heads)
    model = hip.HipModel("dlav1_34", heads, sd)
    x = synth.frames(2, seed=41, h=128, w=128).to(device)
    z = model(x, sigmoid_hm=True)
    det_ref = hip.decode_raw(z["hm"].clone(), z["hps"], z["wh"], z["hm_hp"].clone(), None, z["scale"], None, z["reg"],
                             z["hp_offset"], None, None, K=100, rep_mode=1)
    side = torch.cuda.Stream(device=device)
    with torch.cuda.stream(side):
        outs, det = model.detect(x, graph=False)
        side.synchronize()
        assert torch.equal(det, det_ref)
        for k in heads:
            assert torch.equal(outs[k], z[k])
        outs, det = model.detect(x, graph=True)      # capture + first replay
        side.synchronize()
        assert torch.equal(det, det_ref)
        x2 = synth.frames(2, seed=42, h=128, w=128).to(device)
        z2 = model(x2, sigmoid_hm=True)
        det2_ref = hip.decode_raw(z2["hm"].clone(), z2["hps"], z2["wh"], z2["hm_hp"].clone(), None, z2["scale"], None,
                                  z2["reg"], z2["hp_offset"], None, None, K=100, rep_mode=1)
        x.copy_(x2)                                   # new frame into the captured input buffer
        outs, det = model.detect(x, graph=True)      # pure replay
        side.synchronize()
        assert torch.equal(det, det2_ref)
