"""ORACLE (test infrastructure): functional CPU restatement of the reference DLA-34 graph.

Operates directly on a reference-format ``state_dict`` (same key names as the reference
modules register), float32, NCHW, eval-mode BatchNorm.  Each function cites the reference
lines it follows (paths relative to /root/reference/src/lib/models/networks/).

Pinned against the reference's own modules by ``oracle/tools/make_goldens.py`` (run in the
build container, where /root/reference is importable) -> ``tests/golden/backbone_*.npz``.
"""
import torch
import torch.nn.functional as F

from . import dcn as _dcn

BN_EPS = 1e-5  # nn.BatchNorm2d default, pose_dla_dcn.py:41
GN_EPS = 1e-5  # nn.GroupNorm default, GN.py:7


class Ctx:
    """Carries the state dict plus an optional per-conv hook used by the synthetic-weight
    calibration tool (hook(name, kind, tensor) may rescale weights in ``sd`` and return a
    replacement tensor)."""

    def __init__(self, sd, hook=None, dcn_kind="port", taps=None):
        self.sd = sd
        self.hook = hook
        self.dcn_kind = dcn_kind
        self.taps = taps  # optional dict: name -> tensor, filled when not None

    def tap(self, name, t):
        if self.taps is not None:
            self.taps[name] = t
        return t

    def h(self, name, kind, t):
        if self.hook is not None:
            r = self.hook(name, kind, t)
            if r is not None:
                return r
        return t


def _conv(ctx, x, name, stride=1, padding=0, kind="conv"):
    w = ctx.sd[name + ".weight"]
    b = ctx.sd.get(name + ".bias")
    if ctx.hook is None:  # exactly the reference's nn.Conv2d call
        return F.conv2d(x, w, b, stride=stride, padding=padding)
    y = F.conv2d(x, w, None, stride=stride, padding=padding)
    y = ctx.h(name, kind, y)
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y


def _bn(ctx, x, name):
    sd = ctx.sd
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], False, 0.0, BN_EPS)


def _conv_bn(ctx, x, conv, bn, stride=1, padding=0, relu=True):
    y = _bn(ctx, _conv(ctx, x, conv, stride, padding, kind="conv_bn:" + bn), bn)
    return F.relu(y) if relu else y


def basic_block(ctx, x, p, stride, residual=None):
    """pose_dla_dcn.py:48-62"""
    if residual is None:
        residual = x
    out = _conv_bn(ctx, x, p + ".conv1", p + ".bn1", stride, 1, relu=True)
    out = _conv_bn(ctx, out, p + ".conv2", p + ".bn2", 1, 1, relu=False)
    out = out + residual
    return ctx.tap(p, F.relu(out))


def root(ctx, p, *xs):
    """pose_dla_dcn.py:160-168 (residual_root=False for dla34, :340-343)"""
    y = _conv_bn(ctx, torch.cat(xs, 1), p + ".conv", p + ".bn", 1, 0, relu=True)
    return ctx.tap(p, y)


def tree(ctx, x, p, levels, cin, cout, stride, level_root, children=None):
    """pose_dla_dcn.py:211-224.  The ``residual`` argument of the reference is always
    recomputed inside (:214) so it is not a parameter here."""
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x
    has_project = cin != cout
    if level_root:
        children.append(bottom)
    if levels == 1:
        residual = _conv_bn(ctx, bottom, p + ".project.0", p + ".project.1", 1, 0, relu=False) \
            if has_project else bottom
        x1 = basic_block(ctx, x, p + ".tree1", stride, residual)
        x2 = basic_block(ctx, x1, p + ".tree2", 1)
        return root(ctx, p + ".root", x2, x1, *children)
    # levels == 2: the outer project's result is discarded by the inner tree (dead compute)
    x1 = tree(ctx, x, p + ".tree1", levels - 1, cin, cout, stride, False)
    children.append(x1)
    return tree(ctx, x1, p + ".tree2", levels - 1, cout, cout, 1, False, children=children)


def dla34_base(ctx, x, pre_img=None, pre_hm=None, pre_hm_hp=None):
    """DLA.forward pose_dla_dcn.py:310-322 with dla34 = levels [1,1,1,2,2,1], channels
    [16,32,64,128,256,512] (:340-343)."""
    ch = [16, 32, 64, 128, 256, 512]
    y = []
    x0 = _conv_bn(ctx, x, "base.base_layer.0", "base.base_layer.1", 1, 3)
    if pre_img is not None:
        x0 = x0 + _conv_bn(ctx, pre_img, "base.pre_img_layer.0", "base.pre_img_layer.1", 1, 3)
    if pre_hm is not None:
        x0 = x0 + _conv_bn(ctx, pre_hm, "base.pre_hm_layer.0", "base.pre_hm_layer.1", 1, 3)
    if pre_hm_hp is not None:
        x0 = x0 + _conv_bn(ctx, pre_hm_hp, "base.pre_hm_hp_layer.0", "base.pre_hm_hp_layer.1", 1, 3)
    x = ctx.tap("base.base_layer", x0)
    x = ctx.tap("base.level0", _conv_bn(ctx, x, "base.level0.0", "base.level0.1", 1, 1)); y.append(x)
    x = ctx.tap("base.level1", _conv_bn(ctx, x, "base.level1.0", "base.level1.1", 2, 1)); y.append(x)
    x = tree(ctx, x, "base.level2", 1, ch[1], ch[2], 2, False); y.append(x)
    x = tree(ctx, x, "base.level3", 2, ch[2], ch[3], 2, True); y.append(x)
    x = tree(ctx, x, "base.level4", 2, ch[3], ch[4], 2, True); y.append(x)
    x = tree(ctx, x, "base.level5", 1, ch[4], ch[5], 2, True); y.append(x)
    return y


def deform_conv(ctx, x, p):
    """DeformConv.forward pose_dla_dcn.py:386-389 -> DCN.forward DCNv2/dcn_v2.py:118-128."""
    out = _conv(ctx, x, p + ".conv.conv_offset_mask", 1, 1, kind="offset")
    # chunk(3)+cat(o1,o2) is the identity on the first 18 channels (dcn_v2.py:120-121)
    offset = out[:, :18].contiguous()
    mask = torch.sigmoid(out[:, 18:27]).contiguous()
    w = ctx.sd[p + ".conv.weight"]
    b = ctx.sd[p + ".conv.bias"]
    if ctx.hook is None:
        y = _dcn.dcn_v2_forward(x, w, b, offset, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1, kind=ctx.dcn_kind)
    else:  # calibration path: let the hook see (and rescale) the bias-free contraction
        y = _dcn.dcn_v2_forward(x, w, torch.zeros_like(b), offset, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1,
                                kind=ctx.dcn_kind)
        y = ctx.h(p + ".conv", "conv_bn:" + p + ".actf.0", y)
        y = y + b.view(1, -1, 1, 1)
    y = F.relu(_bn(ctx, y, p + ".actf.0"))
    return ctx.tap(p, y)


def _up(ctx, x, name, f):
    """Depth-wise ConvTranspose2d(o, o, 2f, stride=f, padding=f//2, groups=o), pose_dla_dcn.py:402-404"""
    w = ctx.sd[name + ".weight"]
    return F.conv_transpose2d(x, w, None, stride=f, padding=f // 2, groups=w.shape[0])


def ida_up(ctx, layers, p, startp, endp, up_f):
    """IDAUp.forward pose_dla_dcn.py:411-417 (mutates ``layers`` in place like the reference)."""
    for i in range(startp + 1, endp):
        k = i - startp
        t = deform_conv(ctx, layers[i], "%s.proj_%d" % (p, k))
        t = ctx.tap("%s.up_%d" % (p, k), _up(ctx, t, "%s.up_%d" % (p, k), up_f[k]))
        layers[i] = deform_conv(ctx, t + layers[i - 1], "%s.node_%d" % (p, k))


def dla_up(ctx, layers):
    """DLAUp.forward pose_dla_dcn.py:437-443 with first_level=2: three IDAUp stages."""
    layers = list(layers)
    out = [layers[-1]]
    up_fs = {0: [1, 2], 1: [1, 2, 2], 2: [1, 2, 2, 2]}
    for i in range(len(layers) - 2 - 1):
        ida_up(ctx, layers, "dla_up.ida_%d" % i, len(layers) - i - 2, len(layers), up_fs[i])
        out.insert(0, layers[-1])
    return out


def conv_gru(ctx, x, steps):
    """ConvGRU.forward convGRU.py:72-94 / ConvGRUCell.forward :32-39; the four b* tensors are
    zeros (:42-46) and are omitted."""
    p = "convGRU.cell0."
    h = torch.zeros_like(x[:, :64])
    outs = []
    for s in range(steps):
        rt = torch.sigmoid(_conv(ctx, x, p + "Wir", 1, 1, "gru") + _conv(ctx, h, p + "Whr", 1, 1, "gruh"))
        zt = torch.sigmoid(_conv(ctx, x, p + "Wiz", 1, 1, "gru") + _conv(ctx, h, p + "Whz", 1, 1, "gruh"))
        nt = torch.tanh(_conv(ctx, x, p + "Win", 1, 1, "gru") + rt * _conv(ctx, h, p + "Whn", 1, 1, "gruh"))
        h = (1 - zt) * nt + zt * h
        outs.append(ctx.tap("convGRU.step%d" % s, h))
    return outs


def head(ctx, x, name, use_gn):
    """Head Sequential pose_dla_dcn.py:491-521: conv3x3(64->head_conv,bias) [GroupNorm(32)] ReLU
    conv1x1(head_conv->classes,bias)."""
    y = _conv(ctx, x, name + ".0", 1, 1, kind="head0")
    if use_gn:
        y = F.group_norm(y, 32, ctx.sd[name + ".1.weight"], ctx.sd[name + ".1.bias"], GN_EPS)
        last = name + ".3"
    else:
        last = name + ".2"
    y = F.relu(y)
    return _conv(ctx, y, last, 1, 0, kind="head1:" + name)


def head_routing(heads, use_gru, tracking_task):
    """pose_dla_dcn.py:542-568: which GRU step feeds which head (None = y[-1])."""
    route = {}
    for hname in heads:
        if not use_gru:
            route[hname] = None
        elif tracking_task:
            if hname in ("tracking", "tracking_hp"):
                route[hname] = 0
            elif hname in ("hm", "wh", "reg"):
                route[hname] = 1
            elif hname in ("hm_hp", "hp_offset", "hps", "hps_uncertainty"):
                route[hname] = 2
            elif hname in ("scale", "scale_uncertainty"):
                route[hname] = 3
        else:
            if hname in ("hm", "wh", "reg"):
                route[hname] = 0
            elif hname in ("hm_hp", "hp_offset", "hps"):
                route[hname] = 1
            elif hname == "scale":
                route[hname] = 2
    return route


@torch.no_grad()
def dlaseg_forward(sd, x, heads, arch="dla", tracking_task=False,
                   pre_img=None, pre_hm=None, pre_hm_hp=None,
                   hook=None, dcn_kind="port", taps=None):
    """DLASeg.forward pose_dla_dcn.py:523-570.  Returns the head dict ``z`` (raw logits)."""
    ctx = Ctx(sd, hook, dcn_kind, taps)
    use_gru = arch == "dlav1"
    ys = dla34_base(ctx, x.float(), pre_img, pre_hm, pre_hm_hp)
    ups = dla_up(ctx, ys)
    y = [ups[0], ups[1], ups[2]]  # last_level - first_level = 3 (clone not needed: functional)
    ida_up(ctx, y, "ida_up", 0, 3, [1, 2, 4])
    feat = ctx.tap("feat", y[-1])
    z = {}
    route = head_routing(heads, use_gru, tracking_task)
    if use_gru:
        gru = conv_gru(ctx, feat, 4 if tracking_task else 3)
    for hname in heads:
        if route.get(hname, None) is None and use_gru:
            continue  # the reference silently skips heads with no route (:545-563)
        src = feat if not use_gru else gru[route[hname]]
        z[hname] = head(ctx, src, hname, use_gru)
    return z
