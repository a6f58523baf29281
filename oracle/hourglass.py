"""ORACLE (test infrastructure only): CPU restatement of the reference's stacked-hourglass forward
(src/lib/models/networks/large_hourglass.py:18-307, ``HourglassNet(heads, 2)``) as a functional walk over a
state dict.  Pinned bit-for-bit against the reference module in the build container
(oracle/tools/make_goldens.py -> tests/golden/backbone_hourglass.npz, tests/test_oracle_pins.py).

Only what the detector consumes is computed: ``model(x)[-1]``, i.e. the heads of the LAST stack
(detectors/object_pose.py:135 takes ``[-1]``); the first stack's heads do not influence it.
"""
import torch
import torch.nn.functional as F

DIMS = [256, 256, 384, 384, 384, 512]   # large_hourglass.py:296-298
MODULES = [2, 2, 2, 2, 2, 4]


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


def convolution(sd, p, x, k, stride=1, with_bn=True):
    """:18-31  conv (bias iff no bn) -> bn -> relu"""
    y = F.conv2d(x, sd[p + ".conv.weight"], sd.get(p + ".conv.bias"), stride, (k - 1) // 2)
    if with_bn:
        y = _bn(sd, p + ".bn", y)
    return F.relu(y)


def residual(sd, p, x, stride=1):
    """:50-78"""
    y = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)))
    y = _bn(sd, p + ".bn2", F.conv2d(y, sd[p + ".conv2.weight"], None, 1, 1))
    if (p + ".skip.0.weight") in sd:
        x = _bn(sd, p + ".skip.1", F.conv2d(x, sd[p + ".skip.0.weight"], None, stride, 0))
    return F.relu(y + x)


def _seq(sd, p, x, n, first_stride=1):
    for i in range(n):
        x = residual(sd, "%s.%d" % (p, i), x, first_stride if i == 0 else 1)
    return x


def kp_module(sd, p, x, n, modules, hook=None):
    """:129-189  up1(x) + upsample(low3(low2(low1(x)))), low1 strides by 2 (make_hg_layer :288-291)"""
    cm, nm = modules[0], modules[1]
    up1 = _seq(sd, p + ".up1", x, cm)
    low1 = _seq(sd, p + ".low1", x, cm, 2)
    low2 = kp_module(sd, p + ".low2", low1, n - 1, modules[1:], hook) if n > 1 else _seq(sd, p + ".low2", low1, nm)
    low3 = _seq(sd, p + ".low3", low2, cm)
    out = up1 + F.interpolate(low3, scale_factor=2)   # nn.Upsample(scale_factor=2): nearest
    if hook is not None:
        hook(p, out)
    return out


def hourglass_forward(sd, x, heads, nstack=2, hook=None):
    """exkp.forward :266-286 -> dict of the last stack's head tensors"""
    inter = residual(sd, "pre.1", convolution(sd, "pre.0", x, 7, 2), 2)
    if hook is not None:
        hook("pre", inter)
    cnv = None
    for k in range(nstack):
        kp = kp_module(sd, "kps.%d" % k, inter, 5, MODULES, hook)
        cnv = convolution(sd, "cnvs.%d" % k, kp, 3)
        if hook is not None:
            hook("cnvs.%d" % k, cnv)
        if k < nstack - 1:
            a = _bn(sd, "inters_.%d.1" % k, F.conv2d(inter, sd["inters_.%d.0.weight" % k]))
            b = _bn(sd, "cnvs_.%d.1" % k, F.conv2d(cnv, sd["cnvs_.%d.0.weight" % k]))
            inter = residual(sd, "inters.%d" % k, F.relu(a + b))
    out = {}
    last = nstack - 1
    for h in heads:
        y = convolution(sd, "%s.%d.0" % (h, last), cnv, 3, with_bn=False)
        out[h] = F.conv2d(y, sd["%s.%d.1.weight" % (h, last)], sd["%s.%d.1.bias" % (h, last)])
    return out
