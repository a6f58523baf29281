"""Bring-up diagnostic: dcn16s vs dcn16p vs dcn16 on the network's layer shapes (same data sequence as tools/dcn_ab.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from centerpose_amd import hip
hip.set_default_precision("f16x3")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
SHAPES = [(64, 64, 128), (128, 64, 64), (128, 128, 64), (256, 128, 32), (256, 256, 32), (256, 64, 32), (512, 256, 16)]
g = torch.Generator().manual_seed(1)
for ci, co, hw in SHAPES:
    x = torch.randn(B, ci, hw, hw, generator=g).cuda()
    w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).cuda()
    bias = torch.randn(co, generator=g).cuda()
    off = (torch.randn(B, 18, hw, hw, generator=g) * 1.5).cuda()
    mask = torch.rand(B, 9, hw, hw, generator=g).cuda()
    outs = {}
    for name, dbg in [("dcn16", 32768)] + [("dcn16p_%d" % i, 65536 | 1048576) for i in range(reps)] + [("dcn16s_%d" % i, 65536 | 2097152) for i in range(reps)]:
        hip.lib().cp_set_debug(dbg)
        outs[name] = hip.dcn_v2_forward(x, w, bias, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1)
        torch.cuda.synchronize()
    hip.lib().cp_set_debug(0)
    ref = outs["dcn16"]
    print("%d->%d @%d" % (ci, co, hw))
    for k, v in outs.items():
        d = (v - ref).abs() / ref.abs().max()
        bad = d > 1e-5
        if k in ("dcn16p_0", "dcn16s_0") or int(bad.sum()):
            print("  %-10s max rel %.2e, bad elements %d of %d" % (k, float(d.max()), int(bad.sum()), bad.numel()))
        if int(bad.sum()):
            idx = bad.nonzero()
            print("     b:", sorted(set(idx[:, 0].tolist()))[:20], "\n     n:", sorted(set(idx[:, 1].tolist()))[:40],
                  "\n     y:", sorted(set(idx[:, 2].tolist())), "\n     x:", sorted(set(idx[:, 3].tolist())))
