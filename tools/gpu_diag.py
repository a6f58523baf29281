"""GPU diagnostic sweep (test infrastructure: imports the oracle; a tool, not a pytest module): runs every kernel family against the CPU oracle and prints one
line per case without stopping at the first mismatch.  Usage on the GPU box:
    python tools/gpu_diag.py [--full | --f16x3 | --decode-only | --pnp-only]      (output is also what gpurun shows in its tail)
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)

from centerpose_amd import hip, synth  # noqa: E402
from oracle import backbone as ob  # noqa: E402
from oracle import dcn as odcn  # noqa: E402

dev = torch.device("cuda:0")
results = []
CONV_TOL = 2e-5


def report(name, got, ref, tol):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    err = (got - ref).abs().max().item() if got.numel() else 0.0
    scale = ref.abs().max().item()
    bad = not (err <= tol * max(1.0, scale)) or not torch.isfinite(got).all()
    results.append((name, err, scale, bad))
    print("%-58s err %.3e  ref_max %.3e  %s" % (name, err, scale, "FAIL" if bad else "ok"), flush=True)
    return not bad


def conv_case(B, H, W, Cin, Cout, k, stride, pad, act=0, res=False, affine=True, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bn = hip.lib().cp_conv2d_workspace_bytes  # noqa
    tile = 16 if Cout <= 16 else 32 if Cout <= 32 else 128 if Cout % 128 == 0 else 64
    affine = affine and (Cout % tile == 0)
    sc = torch.rand(Cout, generator=g) + 0.5 if affine else None
    sh = torch.randn(Cout, generator=g) if affine else None
    y = F.conv2d(x, w, None, stride, pad)
    if affine:
        y = y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    r = None
    if res:
        r = torch.randn(y.shape, generator=g)
        y = y + r
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = torch.sigmoid(y)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
    rg = r.permute(0, 2, 3, 1).contiguous().to(dev) if res else None
    out = hip.conv2d_nhwc(xg, w.to(dev), sc.to(dev) if affine else None, sh.to(dev) if affine else None, rg,
                          stride, pad, act)
    torch.cuda.synchronize()
    name = "conv B%d %dx%d %d->%d k%d s%d p%d act%d res%d aff%d" % (B, H, W, Cin, Cout, k, stride, pad, act, res, affine)
    return report(name, out.permute(0, 3, 1, 2), y, CONV_TOL)


def small_value_case():
    """Activations of magnitude 1e-3 .. 1e-5: the binary16 'lo' halves become subnormal; checks they are not flushed."""
    g = torch.Generator().manual_seed(5)
    for mag in (1e-3, 1e-5):
        x = torch.randn(1, 64, 16, 16, generator=g) * mag
        w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
        y = F.conv2d(x.double(), w.double(), None, 1, 1).float()
        out = hip.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(dev), w.to(dev), None, None, None, 1, 1, 0)
        torch.cuda.synchronize()
        o = out.permute(0, 3, 1, 2).cpu()
        rel = float((o - y).abs().max() / y.abs().max())
        # |x| ~ 1e-5 is below binary16's normal range: informational only (absolute error stays < 1e-7)
        bad = rel > 1e-4 and mag >= 1e-3
        results.append(("small", rel, mag, bad))
        print("%-58s rel err %.3e %s" % ("conv small activations |x|~%g" % mag, rel, "FAIL" if bad else "ok"), flush=True)


def dcn_case(B, C, Co, H, W, off_std, seed=0, kat=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(Co, generator=g)
    off = torch.randn(B, 18, H, W, generator=g) * off_std
    mask = torch.rand(B, 9, H, W, generator=g)
    if kat == "zero_offset":  # check_zero_offset, DCNv2/testcpu.py:32-67 generalised: mask .5 -> 0.5*conv
        off.zero_()
        mask.fill_(0.5)
    ref = odcn.dcn_v2_forward(x, w, b, off, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    out = hip.dcn_v2_forward(x.to(dev), w.to(dev), b.to(dev), off.to(dev), mask.to(dev), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    torch.cuda.synchronize()
    ok = report("dcn B%d C%d->%d %dx%d off_std %.1f %s" % (B, C, Co, H, W, off_std, kat or ""), out, ref, CONV_TOL)
    if kat == "zero_offset":
        ref2 = 0.5 * F.conv2d(x, w, None, 1, 1) + b.view(1, -1, 1, 1)
        report("   ... vs 0.5*conv2d+bias KAT", out, ref2, CONV_TOL)
    return ok


def backbone_case(arch, tracking, B, res, taps_on_fail=True, seed=11):
    heads = synth.HEADS_TRACK if tracking else synth.HEADS_POSE
    sd = synth.make_state_dict(arch, heads, tracking)
    x = synth.frames(B, seed=seed, h=res, w=res)
    kw = {}
    if tracking:
        kw = dict(pre_img=synth.frames(B, seed=seed + 1, h=res, w=res),
                  pre_hm=torch.rand(B, 1, res, res, generator=synth._gen(seed, "pre_hm")) ** 8,
                  pre_hm_hp=torch.rand(B, 8, res, res, generator=synth._gen(seed, "pre_hm_hp")) ** 8)
    taps = {}
    t0 = time.time()
    zo = ob.dlaseg_forward(sd, x, heads, arch=arch.split("_")[0], tracking_task=tracking, taps=taps, **kw)
    t_cpu = time.time() - t0
    model = hip.HipModel(arch, heads, sd, tracking_task=tracking)
    kwg = {k: v.to(dev) for k, v in kw.items()}
    zg = model(x.to(dev), **kwg)
    torch.cuda.synchronize()
    ok = True
    tag = "%s%s B%d %d" % (arch, "+trk" if tracking else "", B, res)
    for k in zo:
        ok &= report("backbone %s head %s" % (tag, k), zg[k], zo[k], 1e-3)
    hm_ok = report("backbone %s sigmoid(hm)" % tag, torch.sigmoid(zg["hm"]), torch.sigmoid(zo["hm"]), 1e-3)
    if (not ok or not hm_ok) and taps_on_fail:
        for name in ["base.base_layer", "base.level0", "base.level1", "base.level2.tree1", "base.level2.tree2",
                     "base.level2.root", "base.level3", "base.level4", "base.level5", "dla_up.ida_0.proj_1",
                     "dla_up.ida_0.node_1", "dla_up.ida_1.node_2", "dla_up.ida_2.node_3", "ida_up.node_1", "feat",
                     "convGRU.step0", "convGRU.step1", "convGRU.step2"]:
            oname = {"base.level2": "base.level2.root", "base.level3": "base.level3.tree2.root",
                     "base.level4": "base.level4.tree2.root", "base.level5": "base.level5.root"}.get(name, name)
            if oname not in taps:
                continue
            _, t = model(x.to(dev), tap=name, **kwg)
            torch.cuda.synchronize()
            report("   tap %s" % name, t, taps[oname], 1e-4)
    print("   (oracle CPU forward %.2fs)" % t_cpu)
    return model, sd


def decode_case(B, sem, tracking=False, seed=317, rep_mode=1, sparse=False):
    from oracle import decode as odec
    d = odec.synth_heads(B, seed=seed, tracking=tracking)
    if sparse:  # fewer than K peaks: most of the map exactly zero -> ties at 0 fall back to index order
        keep = (d["hm"] > 0.9)
        d["hm"] = d["hm"] * keep
        d["hm_hp"] = d["hm_hp"] * (d["hm_hp"] > 0.9)
    o = odec.object_pose_decode(
        d["hm"], d["hps"], wh=d["wh"], kps_displacement_std=d.get("hps_uncertainty"), obj_scale=d["scale"],
        obj_scale_uncertainty=d.get("scale_uncertainty"), reg=d["reg"], hm_hp=d["hm_hp"], hp_offset=d["hp_offset"],
        tracking=d.get("tracking"), tracking_hp=d.get("tracking_hp"), K=100, rep_mode=rep_mode,
        tracking_task=tracking, mask_semantics=sem)
    g = {k: torch.from_numpy(v).to(dev) for k, v in d.items()}
    det = hip.decode_raw(g["hm"], g["hps"], g["wh"], g["hm_hp"], g.get("hps_uncertainty"), g["scale"],
                         g.get("scale_uncertainty"), g["reg"], g["hp_offset"], g.get("tracking"), g.get("tracking_hp"),
                         K=100, rep_mode=rep_mode, fit_gaussian=tracking, balance=2.0,
                         legacy_bool_mask=(sem == "bool"))
    torch.cuda.synchronize()
    r = hip.split_detections(det.cpu())
    tag = "decode B%d %s trk%d rep%d%s" % (B, sem, tracking, rep_mode, " sparse" if sparse else "")
    for k in o:
        tol = 2e-6 if k in ("kps_displacement_std", "obj_scale_uncertainty") or tracking else 0.0
        exact = bool((r[k].numpy() == o[k]).all())
        report("%s %s%s" % (tag, k, " [bit-exact]" if exact else ""), r[k], torch.from_numpy(o[k]), tol)


def pnp_case(N=256, noise=0.0, seed=0, npts=16, drop=0.0):
    import numpy as np
    from oracle import pnp as opnp
    rng = np.random.RandomState(seed)
    K = np.array([[663.0287679036459, 0, 300.2775065104167], [0, 663.0287679036459, 395.00066121419275], [0, 0, 1]])
    pts = np.zeros((N, npts, 2), np.float32)
    scale = np.zeros((N, 3), np.float32)
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
            if npts == 16:
                dead[0::2] = False  # displacement points are always present (rep_mode 1)
            p[dead] = -10000
        pts[i] = p
        scale[i] = sc
    cam = np.tile(np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]), (N, 1))
    t0 = time.time()
    out = hip.pnp_solve(torch.from_numpy(pts).to(dev), torch.from_numpy(scale).to(dev), torch.from_numpy(cam).to(dev))
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    out = out.cpu().numpy()
    ref = np.zeros_like(out)
    t0 = time.time()
    for i in range(N):
        try:
            s = opnp.solve_cuboid_pnp(pts[i].astype(np.float64), scale[i].astype(np.float64), K, opencv_return=True)
            s2 = opnp.solve_cuboid_pnp(pts[i].astype(np.float64), scale[i].astype(np.float64), K, opencv_return=False)
        except NotImplementedError:
            continue
        if s is None:
            continue
        ref[i, 0] = 1
        ref[i, 1:4] = s["rvec"]; ref[i, 4:7] = s["tvec"]; ref[i, 7] = s["reproj_err"]
        ref[i, 8:24] = s["projected_points"].reshape(-1)
        ref[i, 24:28] = s["quaternion_xyzw"]; ref[i, 28:31] = s2["location"]; ref[i, 31:35] = s2["quaternion_xyzw"]
    t_cpu = time.time() - t0
    okm = ref[:, 0] == 1
    tag = "pnp N%d npts%d noise %.1f drop %.1f" % (N, npts, noise, drop)
    report(tag + " status", torch.from_numpy((out[:, 0] == 1).astype(np.float32)), torch.from_numpy(okm.astype(np.float32)), 0)
    # rotation geodesic (deg) and relative translation error vs the float64 oracle
    ang = np.zeros(N)
    for i in np.where(okm)[0]:
        Ra = opnp.rodrigues_to_matrix(out[i, 1:4]); Rb = opnp.rodrigues_to_matrix(ref[i, 1:4])
        ang[i] = np.degrees(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))
    report(tag + " rot geodesic deg (tol 1e-3)", torch.from_numpy(ang), torch.zeros(N, dtype=torch.float64), 1e-3)
    rel_t = np.linalg.norm(out[okm, 4:7] - ref[okm, 4:7], axis=1) / np.linalg.norm(ref[okm, 4:7], axis=1)
    report(tag + " rel |dt| (tol 1e-5)", torch.from_numpy(rel_t), torch.zeros(len(rel_t), dtype=torch.float64), 1e-5)
    report(tag + " projected px", torch.from_numpy(out[okm, 8:24]), torch.from_numpy(ref[okm, 8:24]), 1e-6)
    qs = np.sign(np.sum(out[okm, 24:28] * ref[okm, 24:28], axis=1, keepdims=True))
    report(tag + " quat cv", torch.from_numpy(out[okm, 24:28] * qs), torch.from_numpy(ref[okm, 24:28]), 1e-6)
    report(tag + " loc gl", torch.from_numpy(out[okm, 28:31]), torch.from_numpy(ref[okm, 28:31]), 1e-5)
    qs = np.sign(np.sum(out[okm, 31:35] * ref[okm, 31:35], axis=1, keepdims=True))
    report(tag + " quat gl", torch.from_numpy(out[okm, 31:35] * qs), torch.from_numpy(ref[okm, 31:35]), 1e-6)
    print("   gpu %.2f ms (first call, incl. launch)  oracle %.1f ms  LM iters mean %.1f max %d" % (
        t_gpu * 1e3, t_cpu * 1e3, out[okm, 36].mean(), out[okm, 36].max()))


def timing(model, B, res=512, iters=5):
    x = synth.frames(min(B, 4), seed=3, h=res, w=res).to(dev)
    x = x.repeat((B + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:B].contiguous()
    for _ in range(2):
        model(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        model(x)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / iters
    print("timing %s B%d %dx%d: %.2f ms/batch  %.1f img/s  (ws %.2f GB)" % (
        model.arch, B, res, res, dt * 1e3, B / dt, model.workspace_bytes(B, res, res) / 2 ** 30), flush=True)


def main():
    full = "--full" in sys.argv
    global CONV_TOL
    if "--f16x3" in sys.argv:
        hip.set_default_precision("f16x3")
        CONV_TOL = 2e-5  # split-binary16 products: < 2^-20 relative each
        print("precision: f16x3")
    print(hip.lib().cp_version().decode(), torch.cuda.get_device_name(0), flush=True)
    # --- implicit-GEMM conv: every tile config, strides, kernel sizes, ragged M, residual, acts ---
    conv_case(1, 8, 8, 16, 16, 3, 1, 1)             # FRAG16 path, tiny (ragged M)
    conv_case(2, 16, 16, 4, 16, 7, 1, 3, act=1)     # stem-like: Cin=4, K=196 -> padded K
    conv_case(1, 32, 32, 16, 32, 3, 2, 1, act=1)    # BN=32, stride 2
    conv_case(2, 16, 16, 32, 64, 3, 2, 1, act=1)    # BN=64 stride 2
    conv_case(2, 16, 16, 64, 64, 3, 1, 1, act=1, res=True)
    conv_case(1, 16, 16, 64, 128, 3, 1, 1, act=1, res=True)   # BN=128
    conv_case(1, 8, 8, 128, 256, 1, 1, 0)           # 1x1
    conv_case(3, 12, 20, 48, 64, 3, 1, 1, act=2)    # non-pow2 dims, sigmoid
    conv_case(1, 16, 16, 64, 192, 3, 1, 1, affine=True)  # GRU-like N=192
    conv_case(1, 16, 16, 64, 27, 3, 1, 1, affine=False)   # offset-conv-like N=27 (padded to 32)
    conv_case(1, 16, 16, 256, 8, 1, 1, 0, affine=False)   # head-final-like N=8
    conv_case(1, 20, 20, 32, 16, 3, 1, 1, act=1)    # ragged M with FRAG16
    conv_case(2, 16, 16, 32, 64, 3, 1, 1, act=1)    # Cin = 32 (one f16x3 K-step per tap)
    conv_case(1, 16, 16, 128, 128, 3, 1, 1, act=1, res=True)
    conv_case(1, 8, 8, 256, 512, 3, 2, 1)
    conv_case(1, 16, 16, 64, 27, 3, 1, 1, affine=False, seed=3)
    small_value_case()
    # --- DCNv2 ---
    dcn_case(2, 16, 64, 4, 4, 0.0, kat="zero_offset")
    dcn_case(2, 64, 64, 16, 16, 0.0, kat="zero_offset")
    dcn_case(2, 64, 64, 16, 16, 2.0)
    dcn_case(1, 128, 128, 16, 16, 2.0, seed=1)
    dcn_case(1, 256, 128, 8, 8, 5.0, seed=2)       # large offsets: many out-of-image samples
    dcn_case(1, 64, 64, 32, 48, 1.0, seed=3)
    if "--pnp-only" in sys.argv:
        pnp_case(256, 0.0)
        pnp_case(256, 1.0, seed=1)
        pnp_case(256, 1.0, seed=2, drop=0.3)
        pnp_case(128, 0.5, seed=3, npts=8)
        pnp_case(4096, 1.0, seed=4)
        nbad = sum(1 for r in results if r[3])
        print("SUMMARY: %d cases, %d FAIL" % (len(results), nbad))
        return 1 if nbad else 0
    # --- decode vs oracle ---
    decode_case(2, "uint8")
    decode_case(1, "bool")
    decode_case(1, "uint8", rep_mode=0, seed=318)
    decode_case(1, "uint8", rep_mode=4, seed=319)
    decode_case(1, "uint8", sparse=True, seed=320)
    decode_case(1, "uint8", tracking=True)
    if "--decode-only" in sys.argv:
        nbad = sum(1 for r in results if r[3])
        print("SUMMARY: %d cases, %d FAIL" % (len(results), nbad))
        return 1 if nbad else 0
    # --- full network vs oracle ---
    m, _ = backbone_case("dla_34", False, 2, 128)
    backbone_case("dlav1_34", False, 2, 128)
    backbone_case("dla_34", True, 1, 128)
    backbone_case("dlav1_34", True, 1, 128)
    if full:
        m, _ = backbone_case("dla_34", False, 1, 512)
        m1, _ = backbone_case("dlav1_34", False, 1, 512)
        for B in (1, 8, 32):
            timing(m, B)
        timing(m1, 32)
    nbad = sum(1 for r in results if r[3])
    print("SUMMARY: %d cases, %d FAIL" % (len(results), nbad))
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
