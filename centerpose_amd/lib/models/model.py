"""Model factory / checkpoint I/O with the reference's signatures (models/model.py:16-105).

``create_model(arch, heads, head_conv, opt)`` returns a ``HipPoseNet``: it owns the parameters under the
reference's state_dict names and, once moved to the HIP device, runs the whole backbone + heads in
libcenterpose_hip.so.  ``load_model`` accepts the reference's checkpoints unchanged
({'epoch', 'state_dict'[, 'optimizer']}, optional ``module.`` prefixes, tolerant of missing / extra keys).
"""
from collections import OrderedDict

import torch

from centerpose_amd import hip as _hip
from centerpose_amd import synth as _synth

_SUPPORTED = ('dla', 'dlav1', 'hourglass')  # model.py:16-23: the other factories are not named by any benchmark config


class HipPoseNet(object):
    """Stand-in for ``DLASeg`` (pose_dla_dcn.py:457-570): same call signature and return value
    (a list with one dict of head tensors)."""

    def __init__(self, arch, num_layers, heads, head_conv, opt=None):
        if arch == 'hourglass':  # get_large_hourglass_net ignores num_layers / head_conv (large_hourglass.py:311-313)
            self.arch = 'hourglass'
        elif num_layers != 34:
            raise NotImplementedError("only DLA-34 is built (BASELINE configs); got %s_%d" % (arch, num_layers))
        else:
            self.arch = "%s_%d" % (arch, num_layers)
        self.heads = OrderedDict(heads)
        self.head_conv = head_conv
        self.opt = opt
        # each previous-frame stem is created from its own flag (pose_dla_dcn.py:253-271)
        self.tracking = tuple(bool(opt is not None and getattr(opt, f, False)) for f in ('pre_img', 'pre_hm', 'pre_hm_hp'))
        self.tracking_task = bool(opt is not None and getattr(opt, 'tracking_task', False))
        self._spec = _synth.param_spec(self.arch, self.heads, self.tracking, head_conv)
        # construction-time values follow the reference's initialisers where they are deterministic
        # (hm* bias -2.19, zeros elsewhere); everything is expected to come from load_model
        self._sd = OrderedDict()
        last = 3 if arch == 'dlav1' else 2
        for k, shape in self._spec.items():
            if k.endswith('num_batches_tracked'):
                self._sd[k] = torch.zeros((), dtype=torch.long)
            elif k.endswith('running_var') or (len(shape) == 1 and k.endswith('.weight')):
                self._sd[k] = torch.ones(shape)
            else:
                self._sd[k] = torch.zeros(shape)
        for h in self.heads:
            if 'hm' in h:
                if arch == 'hourglass':  # large_hourglass.py:246-247, both stacks
                    for k in range(2):
                        self._sd['%s.%d.1.bias' % (h, k)].fill_(-2.19)
                else:
                    self._sd['%s.%d.bias' % (h, last)].fill_(-2.19)
        self.device = torch.device('cpu')
        self.training = False
        self._hip = None

    # ---- nn.Module-like surface used by the reference's callers ----
    def state_dict(self):
        return OrderedDict(self._sd)

    def load_state_dict(self, state_dict, strict=True):
        missing = [k for k in self._sd if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._sd]
        if strict and (missing or unexpected):
            raise RuntimeError("load_state_dict: missing %s unexpected %s" % (missing[:5], unexpected[:5]))
        for k, v in state_dict.items():
            if k in self._sd:
                if tuple(v.shape) != tuple(self._sd[k].shape):
                    raise RuntimeError("size mismatch for %s" % k)
                self._sd[k] = v.detach().cpu().clone()
        self._hip = None
        return missing, unexpected

    def to(self, device):
        self.device = torch.device(device)
        return self

    def cuda(self):
        return self.to('cuda')

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("centerpose_hip is an inference library")
        return self

    def parameters(self):
        return (v for k, v in self._sd.items() if v.is_floating_point() and 'running_' not in k)

    def _engine(self):
        if self._hip is None:
            # arithmetic of the contractions: opt.precision, else $CENTERPOSE_PRECISION, else exact float32
            import os
            prec = getattr(self.opt, 'precision', None) or os.environ.get('CENTERPOSE_PRECISION', 'f32')
            self._hip = _hip.HipModel(self.arch, self.heads, self._sd, tracking_task=self.tracking_task,
                                      head_conv=self.head_conv, precision=prec)
        return self._hip

    def __call__(self, x, pre_img=None, pre_hm=None, pre_hm_hp=None):
        if not x.is_cuda:
            raise RuntimeError("HipPoseNet runs on the HIP device only: move the model and inputs with .to('cuda')")
        z = self._engine().forward(x, pre_img, pre_hm, pre_hm_hp)
        return [dict(z)]

    forward = __call__


def _dla(num_layers, heads, head_conv=256, down_ratio=4, opt=None):
    return HipPoseNet('dla', num_layers, heads, head_conv, opt)


def _dlav1(num_layers, heads, head_conv=256, down_ratio=4, opt=None):
    return HipPoseNet('dlav1', num_layers, heads, head_conv, opt)


def _hourglass(num_layers, heads, head_conv=256, down_ratio=4, opt=None):
    return HipPoseNet('hourglass', num_layers, heads, head_conv, opt)


_model_factory = {'dla': _dla, 'dlav1': _dlav1, 'hourglass': _hourglass}


def create_model(arch, heads, head_conv, opt=None):
    """models/model.py:26-31"""
    num_layers = int(arch[arch.find('_') + 1:]) if '_' in arch else 0
    arch = arch[:arch.find('_')] if '_' in arch else arch
    if arch not in _model_factory:
        raise NotImplementedError("arch %r is not built by centerpose_hip (supported: %s)" % (arch, _SUPPORTED))
    return _model_factory[arch](num_layers=num_layers, heads=heads, head_conv=head_conv, opt=opt)


def load_model(model, model_path, optimizer=None, resume=False, lr=None, lr_step=None):
    """models/model.py:34-87 (the optimizer-resume branch is training-only and not mirrored)."""
    checkpoint = torch.load(model_path, map_location=lambda storage, loc: storage, weights_only=False)
    print('loaded {}, epoch {}'.format(model_path, checkpoint['epoch']))
    state_dict_ = checkpoint['state_dict']
    state_dict = {}
    for k in state_dict_:
        if k.startswith('module') and not k.startswith('module_list'):
            state_dict[k[7:]] = state_dict_[k]
        else:
            state_dict[k] = state_dict_[k]
    model_state_dict = model.state_dict()
    msg = 'If you see this, your model does not fully load the pre-trained weight. Please make sure you have ' \
          'correctly specified --arch xxx or set the correct --num_classes for your own dataset.'
    for k in list(state_dict):
        if k in model_state_dict:
            if state_dict[k].shape != model_state_dict[k].shape:
                print('Skip loading parameter {}, required shape{}, loaded shape{}. {}'.format(
                    k, model_state_dict[k].shape, state_dict[k].shape, msg))
                state_dict[k] = model_state_dict[k]
        elif not k.startswith('base.fc.'):  # the ImageNet classifier rides along in reference checkpoints
            print('Drop parameter {}.'.format(k) + msg)
    for k in model_state_dict:
        if k not in state_dict:
            print('No param {}.'.format(k) + msg)
            state_dict[k] = model_state_dict[k]
    model.load_state_dict(state_dict, strict=False)
    if optimizer is not None:
        raise NotImplementedError("optimizer resume is training-only")
    return model


def save_model(path, epoch, model, optimizer=None):
    """models/model.py:90-105 (same container so the reference can read it back)."""
    data = {'epoch': epoch, 'state_dict': model.state_dict()}
    if optimizer is not None:
        data['optimizer'] = optimizer.state_dict()
    torch.save(data, path, _use_new_zipfile_serialization=False)
