"""Stress one DCN kernel for rare wrong outputs; dump the failing patches for offline analysis.
usage: python tools/probe/dcn16p_race.py [kernel dcn16p|dcn16s] [iters] [B] [ci] [co] [hw]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from centerpose_amd import hip
hip.set_default_precision("f16x3")
kern = sys.argv[1] if len(sys.argv) > 1 else "dcn16p"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
B, ci, co, hw = [int(v) for v in (sys.argv[3:7] + ["16", "64", "64", "128"][len(sys.argv) - 3 if len(sys.argv) > 3 else 0:])][:4]
g = torch.Generator().manual_seed(5)
x = torch.randn(B, ci, hw, hw, generator=g).cuda()
w = (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).cuda()
bias = torch.randn(co, generator=g).cuda()
off = (torch.randn(B, 18, hw, hw, generator=g) * 1.5).cuda()
mask = torch.rand(B, 9, hw, hw, generator=g).cuda()
dbg = {"dcn16p": 65536 | 1048576, "dcn16s": 65536 | 2097152, "dcn16": 32768}[kern]
hip.lib().cp_set_debug(32768)
ref = hip.dcn_v2_forward(x, w, bias, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1)
torch.cuda.synchronize()
hip.lib().cp_set_debug(dbg)
scale = float(ref.abs().max())
fails = []
nfail = 0
poison = torch.full_like(ref, 12345.0)
for it in range(iters):
    y = hip.dcn_v2_forward(x, w, bias, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    bad = ((y - ref).abs() / scale) > 1e-5
    nb = int(bad.sum())
    if nb:
        nfail += 1
        idx = bad.nonzero()
        b0, y0, x0 = int(idx[0, 0]), int(idx[:, 2].min()) // 8 * 8, int(idx[:, 3].min()) // 16 * 16
        print("iter %d: %d bad; b %s n %d..%d y %s x %d..%d" % (it, nb, sorted(set(idx[:, 0].tolist())), int(idx[:, 1].min()),
              int(idx[:, 1].max()), sorted(set(idx[:, 2].tolist())), int(idx[:, 3].min()), int(idx[:, 3].max())), flush=True)
        if len(fails) < 12:
            fails.append(dict(it=it, b=b0, y0=y0, x0=x0, got=y[b0, :, y0:y0 + 8, x0:x0 + 16].cpu().numpy(),
                              ref=ref[b0, :, y0:y0 + 8, x0:x0 + 16].cpu().numpy()))
    del y
hip.lib().cp_set_debug(0)
if hasattr(hip.lib(), "cp_debug_read_dcn_chk"):
    import ctypes, struct
    buf = (ctypes.c_uint * (8 + 64 * 8))()
    hip.lib().cp_debug_read_dcn_chk(buf)
    print("set-up self-check: %d disagreements logged; candidates matched: h=hb+w_im %d | w=wb+dy %d | w=wb+h_im %d | w=wb+dx[t-1] %d | "
          "w=wb+dx[t+1] %d | w=wb+dx+dy %d" % tuple(buf[0:7]))
    f = lambda u: struct.unpack("f", struct.pack("I", u))[0]
    for k in range(min(int(buf[0]), 64)):
        o = [int(v) for v in buf[8 + 8 * k: 16 + 8 * k]]
        tid = o[1] & 0xffff
        print("   block %d wave %d lane %d tap %d: addr %d expected %d, under the hypothesis %d (%s); w1 %.6g expected %.6g" % (
            o[0], tid >> 6, tid & 63, o[2], o[3], o[4], o[7], "match mask %d" % (o[1] >> 16), f(o[5]), f(o[6])))
print("%s: %d failing launches of %d (B %d %d->%d @%d)" % (kern, nfail, iters, B, ci, co, hw))
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/race_%s.npz" % kern, n=len(fails), **{"%s_%d" % (k, i): np.asarray(f[k]) for i, f in enumerate(fails) for k in f})
