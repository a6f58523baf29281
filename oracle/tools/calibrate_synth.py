"""ORACLE tooling: compute the per-layer calibration factors of centerpose_amd/synth_scales.json.

LSUV-style pass: walk the oracle forward on one seeded batch and
rescale each conv weight so that
  * the input of every BatchNorm has unit std (conv+BN layers, DCN main contraction),
  * DCN offset/mask logits have std 1.5  (offsets O(1-3 px), masks != 0.5),
  * ConvGRU gate pre-activations have std 1,
  * head hidden activations have std 1; heat-map logits std 2.0 (around the -2.19 bias),
    regression heads std 1.
Run once, offline:  python -m oracle.tools.calibrate_synth
"""
import json
import os
import sys

import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from centerpose_amd import synth  # noqa: E402
from oracle import backbone as ob  # noqa: E402

TARGET = {"offset": 1.5, "gru": 1.0, "gruh": 1.0, "head0": 1.0}


def calibrate(arch, tracking, res=256, batch=2, seed=synth.DEFAULT_SEED):
    heads = synth.HEADS_TRACK if tracking else synth.HEADS_POSE
    sd = synth.make_state_dict(arch, heads, tracking, seed=seed, scales={})
    scales = {}

    def hook(name, kind, y):
        wname = name + ".weight"
        if wname in scales:
            return None
        std = float(y.std())
        if std < 1e-12:
            return None  # e.g. hidden-side GRU convs at step 0 (h = 0): calibrate at next step
        if kind.startswith("conv_bn"):
            tgt = 1.0
        elif kind.startswith("head1:"):
            tgt = 2.0 if "hm" in kind else 1.0
        else:
            tgt = TARGET[kind]
        s = tgt / std
        sd[wname].mul_(s)
        scales[wname] = s
        return y * s

    x = synth.frames(batch, seed=seed, h=res, w=res)
    kw = {}
    if tracking:
        kw = dict(pre_img=synth.frames(batch, seed=seed + 1, h=res, w=res),
                  pre_hm=torch.rand(batch, 1, res, res, generator=synth._gen(seed, "pre_hm")) ** 8,
                  pre_hm_hp=torch.rand(batch, 8, res, res, generator=synth._gen(seed, "pre_hm_hp")) ** 8)
    ob.dlaseg_forward(sd, x, heads, arch=arch.split("_")[0], tracking_task=tracking, hook=hook, **kw)
    return scales


def main():
    out = {}
    for arch, tr in (("dla_34", False), ("dlav1_34", False), ("dla_34", True), ("dlav1_34", True)):
        sc = calibrate(arch, tr)
        out[synth.config_key(arch, tr)] = {k: float("%.9g" % v) for k, v in sc.items()}
        print(arch, tr, len(sc), "factors; min %.3g max %.3g" % (min(sc.values()), max(sc.values())))
    path = os.path.join(REPO, "centerpose_amd", "synth_scales.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
