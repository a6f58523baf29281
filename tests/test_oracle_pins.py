"""CPU tests: the oracle restatements against the committed golden vectors, which are outputs of
the REFERENCE's own code (oracle/tools/make_goldens.py).  Where /root/reference is present (build
container) the oracle is additionally compared with the reference function directly."""
import json
import os

import numpy as np
import pytest
import torch

from centerpose_amd import synth
from oracle import backbone as ob
from oracle import dcn as odcn
from oracle import decode as odec
from oracle.tools import make_goldens as mg
from oracle.tools import ref_harness as rh

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CONFIGS = [("dla_34", False), ("dlav1_34", False), ("dla_34", True), ("dlav1_34", True)]


def test_param_spec_matches_reference_state_dict():
    with open(os.path.join(GOLD, "state_dict_keys.json")) as f:
        keys = json.load(f)
    for arch, tr in CONFIGS:
        spec = synth.param_spec(arch, None, tr)
        ref = {k: tuple(v) for k, v in keys[synth.config_key(arch, tr)].items()}
        assert dict(spec) == ref


@pytest.mark.parametrize("arch,tracking", CONFIGS)
def test_backbone_oracle_vs_reference_golden(arch, tracking):
    heads = synth.HEADS_TRACK if tracking else synth.HEADS_POSE
    gold = np.load(os.path.join(GOLD, "backbone_%s.npz" % synth.config_key(arch, tracking)))
    sd = synth.make_state_dict(arch, heads, tracking)
    chk = float(sum(v.double().sum() for v in sd.values() if v.is_floating_point()))
    assert abs(chk - float(gold["_weights_checksum"][0])) < 1e-6 * max(1.0, abs(chk)), "seeded weights differ"
    x, kw = mg.backbone_inputs(tracking)
    z = ob.dlaseg_forward(sd, x, heads, arch=arch.split("_")[0], tracking_task=tracking, **kw)
    for k in heads:
        # bit-exact where the CPU/BLAS build matches the one that generated the fixture; 1e-5 otherwise
        np.testing.assert_allclose(z[k].numpy(), gold[k], rtol=0, atol=1e-5, err_msg=k)


def test_hourglass_spec_and_oracle_vs_reference_golden():
    """Stacked hourglass (large_hourglass.py): parameter names / shapes / order and the oracle's forward against the
    reference module's own output on the seeded weights."""
    from oracle import hourglass as oh

    with open(os.path.join(GOLD, "state_dict_keys.json")) as f:
        keys = json.load(f)["hourglass"]
    spec = synth.param_spec("hourglass", synth.HEADS_POSE)
    assert {k: tuple(v) for k, v in keys.items()} == dict(spec)
    gold = np.load(os.path.join(GOLD, "backbone_hourglass.npz"))
    sd = synth.make_state_dict("hourglass", synth.HEADS_POSE)
    chk = float(sum(v.double().sum() for v in sd.values() if v.is_floating_point()))
    assert abs(chk - float(gold["_weights_checksum"][0])) < 1e-6 * max(1.0, abs(chk)), "seeded weights differ"
    x, _ = mg.backbone_inputs(False)
    z = oh.hourglass_forward(sd, x, synth.HEADS_POSE)
    for k in synth.HEADS_POSE:
        np.testing.assert_allclose(z[k].numpy(), gold[k], rtol=0, atol=1e-5, err_msg=k)
    if rh.available():
        model = rh.create_reference_model("hourglass", synth.HEADS_POSE)
        model.load_state_dict(sd, strict=True)
        with torch.no_grad():
            zr = model(x)[-1]
        for k in synth.HEADS_POSE:
            assert torch.equal(zr[k], z[k]), k


def test_dcn_oracle_vs_reference_golden_and_kat():
    gold = np.load(os.path.join(GOLD, "dcn_ref.npz"))
    x, w, b, off, mask = mg.dcn_case()
    y = odcn.dcn_v2_forward(x, w, b, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1, kind="port")
    np.testing.assert_allclose(y.numpy(), gold["y"], rtol=0, atol=2e-6)
    # independent float64 restatement bounds the float32 oracle's own rounding
    y64 = odcn.dcn_v2_forward_f64(x, w, b, off, mask)
    assert float((y.double() - y64).abs().max()) < 2e-5
    # the reference's known-answer test (DCNv2/testcpu.py:32-67): identity weights, mask .5 => 2*out == in
    xi = torch.from_numpy(gold["kat_in"])
    wi = torch.zeros(2, 2, 3, 3)
    wi[0, 0, 1, 1] = 1.0
    wi[1, 1, 1, 1] = 1.0
    yi = odcn.dcn_v2_forward(xi, wi, torch.zeros(2), torch.zeros(2, 18, 4, 4), torch.full((2, 9, 4, 4), 0.5),
                             3, 3, 1, 1, 1, 1, 1, 1, 1, kind="port")
    assert float((yi * 2 - xi).abs().max()) < 1e-10
    np.testing.assert_array_equal(yi.numpy(), gold["kat_out"])


@pytest.mark.skipif(not odcn.have_reference(), reason="oracle/_ref not built (no /root/reference)")
def test_dcn_im2col_port_bit_exact_vs_reference_binary():
    x, w, b, off, mask = mg.dcn_case(seed=9, B=1, C=8, Co=4, H=9, W=7)
    off = off * 3  # many samples fall outside the image
    a, _, _ = odcn.im2col(x, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1, kind="port")
    r, _, _ = odcn.im2col(x, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1, kind="reference")
    assert torch.equal(a, r)


@pytest.mark.parametrize("i", range(len(mg.DCN_GENERIC_SHAPES)))
def test_dcn_generic_shapes_port_vs_reference(i):
    """deformable groups / strides / dilations / other kernel sizes: the C port against the golden written by the
    reference's compiled CPU source (tests/golden/dcn_generic_ref.npz) and, where oracle/_ref exists, bit for bit
    against that binary's im2col."""
    gold = np.load(os.path.join(GOLD, "dcn_generic_ref.npz"))
    x, w, b, off, mask, args = mg.dcn_generic_case(i)
    y = odcn.dcn_v2_forward(x, w, b, off, mask, *args, kind="port")
    np.testing.assert_allclose(y.numpy(), gold["y%d" % i], rtol=0, atol=2e-6)
    if odcn.have_reference():
        kh, kw, sh, sw, ph, pw, dh, dw, dg = args
        a, _, _ = odcn.im2col(x, off, mask, kh, kw, ph, pw, sh, sw, dh, dw, dg, kind="port")
        r, _, _ = odcn.im2col(x, off, mask, kh, kw, ph, pw, sh, sw, dh, dw, dg, kind="reference")
        assert torch.equal(a, r)


def _decode_oracle(d, tracking, sem, rep_mode=1):
    return odec.object_pose_decode(
        d["hm"], d["hps"], wh=d["wh"], kps_displacement_std=d.get("hps_uncertainty"), obj_scale=d["scale"],
        obj_scale_uncertainty=d.get("scale_uncertainty"), reg=d["reg"], hm_hp=d["hm_hp"], hp_offset=d["hp_offset"],
        tracking=d.get("tracking"), tracking_hp=d.get("tracking_hp"), K=100, rep_mode=rep_mode,
        tracking_task=tracking, mask_semantics=sem)


@pytest.mark.parametrize("name,tracking,sem,B,seed,rep", [
    ("decode_pose_bool", False, "bool", 2, 317, 1),
    ("decode_pose_uint8", False, "uint8", 2, 317, 1),
    ("decode_track_uint8", True, "uint8", 1, 317, 1),
    ("decode_pose_uint8_rep0", False, "uint8", 1, 318, 0),
])
def test_decode_oracle_vs_reference_golden(name, tracking, sem, B, seed, rep):
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    d = odec.synth_heads(B, seed=seed, tracking=tracking)
    o = _decode_oracle(d, tracking, sem, rep)
    assert set(o) == set(gold.files)
    for k in gold.files:
        if k in ("kps_displacement_std", "obj_scale_uncertainty"):  # sqrt(exp()) : libm vs torch ulp
            np.testing.assert_allclose(o[k], gold[k], rtol=2e-6, atol=0, err_msg=k)
        else:
            np.testing.assert_array_equal(o[k], gold[k], err_msg=k)
    if sem == "uint8" and rep == 1:
        assert (o["kps_heatmap_mean"] != -10000).mean() > 0.5  # the filter is not degenerate


@pytest.mark.skipif(not rh.available(), reason="/root/reference not present")
def test_decode_oracle_vs_reference_live():
    d = odec.synth_heads(1, seed=4242)
    for sem in ("bool", "uint8"):
        r = mg.reference_decode_run(d, False, sem)
        o = _decode_oracle(d, False, sem)
        for k in r:
            np.testing.assert_array_equal(o[k], r[k], err_msg="%s/%s" % (sem, k))


@pytest.mark.skipif(not rh.available(), reason="/root/reference not present")
def test_backbone_oracle_vs_reference_live_bit_exact():
    arch, tr = "dlav1_34", False
    model = rh.create_reference_model(arch, synth.HEADS_POSE, tr)
    sd = synth.make_state_dict(arch, seed=5)
    model.load_state_dict(sd, strict=True)
    x = synth.frames(1, seed=6, h=96, w=64)
    with torch.no_grad():
        zr = model(x, None, None, None)[-1]
    zo = ob.dlaseg_forward(sd, x, synth.HEADS_POSE, arch="dlav1")
    for k in zr:
        assert torch.equal(zr[k], zo[k]), k
