"""Seeded synthetic inputs for benchmarking and parity tests (no dataset / checkpoint is
reachable from the build or GPU boxes).

* ``param_spec``      — name -> shape table of the reference ``DLASeg`` state dict
                        (reference: src/lib/models/networks/pose_dla_dcn.py:227-322, 377-521,
                        convGRU.py:7-30, GN.py:4-9).  Pinned against the reference's own
                        ``model.state_dict()`` by tests/golden/state_dict_keys_*.json.
* ``make_state_dict`` — variance-preserving random weights in the reference checkpoint format.
                        Each tensor is drawn from its own generator (seed, crc32(name)) and then
                        multiplied by a per-layer calibration factor from ``synth_scales.json``
                        (computed once, offline, by oracle/tools/calibrate_synth.py so that
                        activations stay O(1), DCN offsets are O(1-3 px) and heat-map logits have
                        std ~2 around the -2.19 bias).
* ``frames``          — uint8-uniform RGB frames, normalised exactly as the reference's
                        ``pre_process`` does (base_detector.py:132; mean/std opts.py:436-437).
"""
import json
import math
import os
import zlib
from collections import OrderedDict

import torch

MEAN = [0.408, 0.447, 0.470]
STD = [0.289, 0.274, 0.278]
DEFAULT_SEED = 317  # the reference's default --seed (opts.py:56)

HEADS_POSE = OrderedDict([("hm", 1), ("wh", 2), ("hps", 16), ("reg", 2), ("hm_hp", 8),
                          ("hp_offset", 2), ("scale", 3)])
HEADS_TRACK = OrderedDict([("hm", 1), ("wh", 2), ("hps", 16), ("hps_uncertainty", 16), ("reg", 2),
                           ("hm_hp", 8), ("hp_offset", 2), ("scale", 3), ("scale_uncertainty", 3),
                           ("tracking", 2), ("tracking_hp", 16)])

_SCALES_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_scales.json")


def config_key(arch, tracking):
    return "%s%s" % (arch.split("_")[0], "_track" if tracking else "")


def _bn(spec, name, c):
    spec[name + ".weight"] = (c,)
    spec[name + ".bias"] = (c,)
    spec[name + ".running_mean"] = (c,)
    spec[name + ".running_var"] = (c,)
    spec[name + ".num_batches_tracked"] = ()


def _block(spec, p, cin, cout):
    spec[p + ".conv1.weight"] = (cout, cin, 3, 3)
    _bn(spec, p + ".bn1", cout)
    spec[p + ".conv2.weight"] = (cout, cout, 3, 3)
    _bn(spec, p + ".bn2", cout)


def _tree(spec, p, levels, cin, cout, level_root, root_dim=0):
    if root_dim == 0:
        root_dim = 2 * cout
    if level_root:
        root_dim += cin
    if levels == 1:
        _block(spec, p + ".tree1", cin, cout)
        _block(spec, p + ".tree2", cout, cout)
        spec[p + ".root.conv.weight"] = (cout, root_dim, 1, 1)
        _bn(spec, p + ".root.bn", cout)
    else:
        _tree(spec, p + ".tree1", levels - 1, cin, cout, False, 0)
        _tree(spec, p + ".tree2", levels - 1, cout, cout, False, root_dim + cout)
    if cin != cout:
        spec[p + ".project.0.weight"] = (cout, cin, 1, 1)
        _bn(spec, p + ".project.1", cout)


def _deform(spec, p, chi, cho):
    _bn(spec, p + ".actf.0", cho)
    spec[p + ".conv.weight"] = (cho, chi, 3, 3)
    spec[p + ".conv.bias"] = (cho,)
    spec[p + ".conv.conv_offset_mask.weight"] = (27, chi, 3, 3)
    spec[p + ".conv.conv_offset_mask.bias"] = (27,)


def _ida(spec, p, o, channels, up_f):
    for i in range(1, len(channels)):
        f = int(up_f[i])
        _deform(spec, "%s.proj_%d" % (p, i), channels[i], o)
        spec["%s.up_%d.weight" % (p, i)] = (o, 1, 2 * f, 2 * f)
        _deform(spec, "%s.node_%d" % (p, i), o, o)


HG_DIMS = [256, 256, 384, 384, 384, 512]   # large_hourglass.py:296-298
HG_MODULES = [2, 2, 2, 2, 2, 4]


def _hg_residual(spec, p, cin, cout, stride=1):
    """large_hourglass.py:50-78: conv1-bn1-relu, conv2-bn2, optional 1x1 skip + bn"""
    spec[p + ".conv1.weight"] = (cout, cin, 3, 3)
    _bn(spec, p + ".bn1", cout)
    spec[p + ".conv2.weight"] = (cout, cout, 3, 3)
    _bn(spec, p + ".bn2", cout)
    if stride != 1 or cin != cout:
        spec[p + ".skip.0.weight"] = (cout, cin, 1, 1)
        _bn(spec, p + ".skip.1", cout)


def _hg_kp(spec, p, n, dims, modules):
    """kp_module (large_hourglass.py:129-189): up1 | low1 (stride 2) | low2 (recursive) | low3 (reversed)"""
    cur, nxt, cm, nm = dims[0], dims[1], modules[0], modules[1]
    for i in range(cm):
        _hg_residual(spec, "%s.up1.%d" % (p, i), cur, cur)
    for i in range(cm):
        _hg_residual(spec, "%s.low1.%d" % (p, i), cur if i == 0 else nxt, nxt, 2 if i == 0 else 1)
    if n > 1:
        _hg_kp(spec, p + ".low2", n - 1, dims[1:], modules[1:])
    else:
        for i in range(nm):
            _hg_residual(spec, "%s.low2.%d" % (p, i), nxt, nxt)
    for i in range(cm):
        _hg_residual(spec, "%s.low3.%d" % (p, i), nxt, nxt if i < cm - 1 else cur)


def hourglass_param_spec(heads=None, nstack=2):
    """OrderedDict name -> shape of the reference ``HourglassNet(heads, 2)`` state dict (large_hourglass.py:191-307)."""
    heads = heads or HEADS_POSE
    s = OrderedDict()
    s["pre.0.conv.weight"] = (128, 3, 7, 7)
    _bn(s, "pre.0.bn", 128)
    _hg_residual(s, "pre.1", 128, 256, 2)
    for k in range(nstack):
        _hg_kp(s, "kps.%d" % k, 5, HG_DIMS, HG_MODULES)
    for k in range(nstack):
        s["cnvs.%d.conv.weight" % k] = (256, 256, 3, 3)
        _bn(s, "cnvs.%d.bn" % k, 256)
    for k in range(nstack - 1):
        _hg_residual(s, "inters.%d" % k, 256, 256)
    for nm in ("inters_", "cnvs_"):
        for k in range(nstack - 1):
            s["%s.%d.0.weight" % (nm, k)] = (256, 256, 1, 1)
            _bn(s, "%s.%d.1" % (nm, k), 256)
    for h, classes in heads.items():
        for k in range(nstack):
            s["%s.%d.0.conv.weight" % (h, k)] = (256, 256, 3, 3)
            s["%s.%d.0.conv.bias" % (h, k)] = (256,)
            s["%s.%d.1.weight" % (h, k)] = (classes, 256, 1, 1)
            s["%s.%d.1.bias" % (h, k)] = (classes,)
    return s


def param_spec(arch="dla_34", heads=None, tracking=False, head_conv=256):
    """OrderedDict name -> shape of the reference state dict for 'dla_34' / 'dlav1_34' (DLASeg) / 'hourglass'."""
    if arch == "hourglass":
        return hourglass_param_spec(heads)
    base_arch = arch.split("_")[0]
    assert base_arch in ("dla", "dlav1"), arch
    heads = heads or (HEADS_TRACK if tracking else HEADS_POSE)
    ch = [16, 32, 64, 128, 256, 512]
    s = OrderedDict()
    s["base.base_layer.0.weight"] = (16, 3, 7, 7)
    _bn(s, "base.base_layer.1", 16)
    s["base.level0.0.weight"] = (16, 16, 3, 3)
    _bn(s, "base.level0.1", 16)
    s["base.level1.0.weight"] = (32, 16, 3, 3)
    _bn(s, "base.level1.1", 32)
    _tree(s, "base.level2", 1, ch[1], ch[2], False)
    _tree(s, "base.level3", 2, ch[2], ch[3], True)
    _tree(s, "base.level4", 2, ch[3], ch[4], True)
    _tree(s, "base.level5", 1, ch[4], ch[5], True)
    # the three previous-frame stems exist independently (pose_dla_dcn.py:253-271: opt.pre_img / pre_hm / pre_hm_hp);
    # `tracking` may be a bool (all or none) or a (pre_img, pre_hm, pre_hm_hp) triple
    pre = tuple(bool(v) for v in tracking) if isinstance(tracking, (tuple, list)) else (bool(tracking),) * 3
    for on, (nm, cin) in zip(pre, (("pre_img_layer", 3), ("pre_hm_layer", 1), ("pre_hm_hp_layer", 8))):
        if on:
            s["base.%s.0.weight" % nm] = (16, cin, 7, 7)
            _bn(s, "base.%s.1" % nm, 16)
    _ida(s, "dla_up.ida_0", 256, [256, 512], [1, 2])
    _ida(s, "dla_up.ida_1", 128, [128, 256, 256], [1, 2, 2])
    _ida(s, "dla_up.ida_2", 64, [64, 128, 128, 128], [1, 2, 2, 2])
    if base_arch == "dlav1":
        for g in ("Wir", "Whr", "Wiz", "Whz", "Win", "Whn"):
            s["convGRU.cell0.%s.weight" % g] = (64, 64, 3, 3)
            if g[1] == "i":
                s["convGRU.cell0.%s.bias" % g] = (64,)
    _ida(s, "ida_up", 64, [64, 128, 256], [1, 2, 4])
    for h, classes in heads.items():
        s[h + ".0.weight"] = (head_conv, 64, 3, 3)
        s[h + ".0.bias"] = (head_conv,)
        last = 2
        if base_arch == "dlav1":
            s[h + ".1.weight"] = (head_conv,)
            s[h + ".1.bias"] = (head_conv,)
            last = 3
        s["%s.%d.weight" % (h, last)] = (classes, head_conv, 1, 1)
        s["%s.%d.bias" % (h, last)] = (classes,)
    return s


def _gen(seed, name):
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    return g


def _bilinear_up(shape):
    """fill_up_weights, pose_dla_dcn.py:365-374"""
    o, _, k, _ = shape
    w = torch.zeros(shape)
    f = math.ceil(k / 2)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    for i in range(k):
        for j in range(k):
            w[0, 0, i, j] = (1 - math.fabs(i / f - c)) * (1 - math.fabs(j / f - c))
    w[1:, 0] = w[0, 0]
    return w


def load_scales(arch, tracking):
    if not os.path.exists(_SCALES_PATH):
        return {}
    with open(_SCALES_PATH) as f:
        allsc = json.load(f)
    return allsc.get(config_key(arch, tracking), {})


def make_state_dict(arch="dla_34", heads=None, tracking=False, seed=DEFAULT_SEED, scales=None,
                    head_conv=256):
    """Random weights in the reference's ``state_dict`` format (float32 CPU tensors)."""
    spec = param_spec(arch, heads, tracking, head_conv)
    if scales is None:
        scales = load_scales(arch, tracking)
    heads_ = heads or (HEADS_TRACK if tracking else HEADS_POSE)
    last = 3 if arch.split("_")[0] == "dlav1" else 2
    final_hm_bias = {"%s.%d.bias" % (h, last) for h in heads_ if "hm" in h}
    if arch == "hourglass":  # heat[-1].bias.fill_(-2.19) on every stack (large_hourglass.py:246-247)
        final_hm_bias = {"%s.%d.1.bias" % (h, k) for h in heads_ if "hm" in h for k in range(2)}
    sd = OrderedDict()
    for name, shape in spec.items():
        g = _gen(seed, name)
        leaf = name.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            t = torch.zeros((), dtype=torch.long)
        elif leaf == "running_mean":
            t = torch.randn(shape, generator=g) * 0.1
        elif leaf == "running_var":
            t = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif len(shape) == 1 and leaf == "weight":  # BN / GN gamma
            t = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif len(shape) == 1:  # biases (BN beta, conv bias)
            if ".up_" in name:
                raise AssertionError(name)
            is_final = name in final_hm_bias
            if is_final:
                t = torch.full(shape, -2.19)  # pose_dla_dcn.py:509-510
            else:
                t = torch.randn(shape, generator=g) * 0.1
        elif ".up_" in name:
            t = _bilinear_up(shape) * (torch.rand(shape, generator=g) * 0.2 + 0.9)
        else:  # conv weight, He-normal on fan_in
            fan_in = shape[1] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
            if arch == "hourglass":
                # ~100 residual / merge additions deep: damp the branch that is ADDED (conv2 of a residual, the skip
                # and merge paths keep unit gain) so activations neither explode nor vanish without a calibration pass
                if name.endswith(".conv2.weight"):
                    t = t * 0.35
                elif len(shape) == 4 and shape[2] == 1 and name.count(".") == 3 and name.split(".")[0] in heads_:
                    t = t * 0.5  # final 1x1 of a head: logits of O(1) around the -2.19 bias
        if name in scales:
            t = t * float(scales[name])
        sd[name] = t.float() if t.dtype != torch.long else t
    return sd


def frames_u8(batch, seed=DEFAULT_SEED, h=512, w=512):
    """[B,h,w,3] uint8 HWC (BGR) frames: the 8-bit images `frames` normalises (what cv2.imread hands BaseDetector.pre_process)."""
    g = _gen(seed, "frames")
    return torch.randint(0, 256, (batch, h, w, 3), generator=g, dtype=torch.uint8)


def frames(batch, seed=DEFAULT_SEED, h=512, w=512, device="cpu"):
    """[B,3,h,w] float32: uint8 uniform noise, BGR mean/std normalisation (base_detector.py:132)."""
    u8 = frames_u8(batch, seed, h, w)
    x = (u8.float() / 255.0 - torch.tensor(MEAN)) / torch.tensor(STD)
    return x.permute(0, 3, 1, 2).contiguous().to(device)


def drawn_heads(batch, seed=DEFAULT_SEED, hw=128):
    """Head tensors drawn directly ("Decode-only (config 2)"): hm / hm_hp = rand()**8 (sparse peaks, top-100 is
    tie-free), hps ~ N(0, 5^2), wh ~ U(5, 35), reg / hp_offset ~ U(0, 1), scale ~ U(0.5, 1.5).  Post-sigmoid, float32 NCHW."""
    g = _gen(seed, "drawn_heads")
    r = lambda c: torch.rand(batch, c, hw, hw, generator=g)
    return {"hm": r(1) ** 8, "hm_hp": r(8) ** 8, "hps": torch.randn(batch, 16, hw, hw, generator=g) * 5.0,
            "wh": r(2) * 30.0 + 5.0, "reg": r(2), "hp_offset": r(2), "scale": r(3) + 0.5}


def rendered_heads(batch, seed=DEFAULT_SEED, n_obj=(1, 10), img=512, hw=128, sigma=1.5, noise=0.02):
    """"Objectron-shaped" head tensors: per image 1-10 cuboids (dataset_combined.py:128 max_objs) of random
    pose and size projected through the demo camera (demo.py:143-144) and the 512 -> 128 affine, written the way the
    reference builds its ground truth (dataset_combined.py:1033-1127): hm = Gaussian at the integer 2-D box centre, wh / reg /
    hps / scale at that pixel, hm_hp[j] = Gaussian at vertex j, hp_offset = sub-pixel rest, + |N(0, noise^2)| background.
    Returns (heads: float32 NCHW tensors with hm / hm_hp post-sigmoid, objects per image)."""
    import numpy as np

    from .lib.utils.pnp.cuboid_objectron import Cuboid3d

    rng = np.random.RandomState(int(seed) % (2 ** 31))
    K = np.array([[663.0287679036459, 0, 300.2775065104167], [0, 663.0287679036459, 395.00066121419275], [0, 0, 1]])
    f32 = np.float32
    h = {"hm": np.zeros((batch, 1, hw, hw), f32), "hm_hp": np.zeros((batch, 8, hw, hw), f32),
         "hps": np.zeros((batch, 16, hw, hw), f32), "wh": np.zeros((batch, 2, hw, hw), f32), "reg": np.zeros((batch, 2, hw, hw), f32),
         "hp_offset": np.zeros((batch, 2, hw, hw), f32), "scale": np.ones((batch, 3, hw, hw), f32)}
    ys, xs = np.mgrid[0:hw, 0:hw]
    ratio = hw / float(img)
    counts = []
    for b in range(batch):
        want = int(rng.randint(n_obj[0], n_obj[1] + 1))
        used, placed, tries = set(), 0, 0
        while placed < want and tries < 400:
            tries += 1
            size = np.array([rng.uniform(0.5, 1.5), 1.0, rng.uniform(0.5, 1.5)]) * rng.uniform(0.12, 0.28)
            q = rng.randn(4)
            q /= np.linalg.norm(q)
            x, y, z, w = q
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            t = np.array([rng.uniform(-0.9, 0.9), rng.uniform(-1.2, 0.4), rng.uniform(2.0, 4.0)])
            V = np.asarray(Cuboid3d(size).get_vertices(), np.float64)
            cam = V @ R.T + t
            if cam[:, 2].min() < 0.3:
                continue
            uv = (cam / cam[:, 2:3]) @ K.T
            uv = uv[:, :2]
            if uv.min() < 8 or uv.max() > img - 8:
                continue
            kp = uv * ratio
            x0, y0, x1, y1 = kp[:, 0].min(), kp[:, 1].min(), kp[:, 0].max(), kp[:, 1].max()
            ct = np.array([(x0 + x1) / 2, (y0 + y1) / 2])
            ci = np.floor(ct).astype(int)
            pix = [tuple(np.floor(k).astype(int)) for k in kp]
            keys = [("c",) + tuple(ci)] + [("k",) + pq for pq in pix]
            # objects apart: centres / vertices never collide (hp_offset is one map shared by all joints)
            if len(set(pix)) < 8 or any((k[0], k[1] + dx, k[2] + dy) in used for k in keys for dx in range(-5, 6) for dy in range(-5, 6)):
                continue
            used.update(keys)
            g = np.exp(-((xs - ci[0]) ** 2 + (ys - ci[1]) ** 2) / (2 * sigma ** 2)).astype(f32)
            h["hm"][b, 0] = np.maximum(h["hm"][b, 0], g * f32(0.95))
            h["wh"][b, :, ci[1], ci[0]] = [x1 - x0, y1 - y0]
            h["reg"][b, :, ci[1], ci[0]] = ct - ci
            h["scale"][b, :, ci[1], ci[0]] = size / size[1]
            for j in range(8):
                h["hps"][b, 2 * j:2 * j + 2, ci[1], ci[0]] = kp[j] - ci
                pj = np.array(pix[j])
                gj = np.exp(-((xs - pj[0]) ** 2 + (ys - pj[1]) ** 2) / (2 * sigma ** 2)).astype(f32)
                h["hm_hp"][b, j] = np.maximum(h["hm_hp"][b, j], gj * f32(0.9))
                h["hp_offset"][b, :, pj[1], pj[0]] = kp[j] - pj
            placed += 1
        counts.append(placed)
    h["hm"] = np.maximum(h["hm"], np.abs(rng.randn(batch, 1, hw, hw) * noise).astype(f32).clip(0, 0.2))
    h["hm_hp"] = np.maximum(h["hm_hp"], np.abs(rng.randn(batch, 8, hw, hw) * noise).astype(f32).clip(0, 0.2))
    return {k: torch.from_numpy(v) for k, v in h.items()}, counts
