"""Test helper: "Objectron-shaped" synthetic head tensors rendered from known cuboid poses, following the
reference's ground-truth construction (SURVEY.md section 8 'Head semantics';
/root/reference/src/lib/datasets/dataset_combined.py:1033-1127): hm = Gaussian at the integer box centre,
wh / reg / hps / scale written at that pixel, hm_hp[j] = Gaussian at vertex j, hp_offset = sub-pixel rest."""
import numpy as np

from oracle import pnp as opnp

K_DEMO = np.array([[663.0287679036459, 0, 300.2775065104167], [0, 663.0287679036459, 395.00066121419275], [0, 0, 1]])


def random_pose(rng, z_range=(2.0, 4.0)):
    q = rng.randn(4)
    R = opnp.quat_xyzw_to_matrix(q / np.linalg.norm(q))
    t = np.array([rng.uniform(-0.25, 0.25), rng.uniform(-0.25, 0.25), rng.uniform(*z_range)])
    return R, t


def render(B, n_obj, seed, K=K_DEMO, img=512, out=128, sigma=1.5):
    """Returns (heads dict of float32 NCHW arrays, post-sigmoid hm/hm_hp; list per image of objects
    dict(R, t, scale, kps_img (8x2)))."""
    rng = np.random.RandomState(seed)
    f32 = np.float32
    H = W = out
    heads = {"hm": np.zeros((B, 1, H, W), f32), "hm_hp": np.zeros((B, 8, H, W), f32),
             "hps": np.zeros((B, 16, H, W), f32), "wh": np.zeros((B, 2, H, W), f32),
             "reg": np.zeros((B, 2, H, W), f32), "hp_offset": np.zeros((B, 2, H, W), f32),
             "scale": np.ones((B, 3, H, W), f32)}
    ys, xs = np.mgrid[0:H, 0:W]
    scenes = []
    ratio = out / float(img)
    for b in range(B):
        objs = []
        used = set()
        tries = 0
        while len(objs) < n_obj and tries < 200:
            tries += 1
            scale = np.array([rng.uniform(0.5, 1.5), 1.0, rng.uniform(0.5, 1.5)]) * rng.uniform(0.15, 0.3)
            R, t = random_pose(rng)
            V = opnp.cuboid_vertices(scale)
            uv = opnp.project_points(V, opnp.matrix_to_rodrigues(R), t, K)
            if uv.min() < 8 or uv.max() > img - 8:
                continue
            kp = uv * ratio
            x0, y0, x1, y1 = kp[:, 0].min(), kp[:, 1].min(), kp[:, 0].max(), kp[:, 1].max()
            ct = np.array([(x0 + x1) / 2, (y0 + y1) / 2])
            ci = np.floor(ct).astype(int)
            pix = [tuple(np.floor(k).astype(int)) for k in kp]
            keys = [("c",) + tuple(ci)] + [("k",) + p for p in pix]
            # keep objects apart so centres / offsets never collide (hp_offset is one map shared by all joints)
            if any((k[0], k[1] + dx, k[2] + dy) in used for k in keys for dx in range(-6, 7) for dy in range(-6, 7)):
                continue
            if len(set(pix)) < 8:
                continue
            for k in keys:
                used.add(k)
            g = np.exp(-((xs - ci[0]) ** 2 + (ys - ci[1]) ** 2) / (2 * sigma ** 2)).astype(f32)
            heads["hm"][b, 0] = np.maximum(heads["hm"][b, 0], g * f32(0.95))
            heads["wh"][b, :, ci[1], ci[0]] = [x1 - x0, y1 - y0]
            heads["reg"][b, :, ci[1], ci[0]] = ct - ci
            heads["scale"][b, :, ci[1], ci[0]] = scale / scale[1] * 0.7  # any positive multiple: PnP uses the ratio
            for j in range(8):
                heads["hps"][b, 2 * j:2 * j + 2, ci[1], ci[0]] = kp[j] - ci
                pj = np.array(pix[j])
                gj = np.exp(-((xs - pj[0]) ** 2 + (ys - pj[1]) ** 2) / (2 * sigma ** 2)).astype(f32)
                heads["hm_hp"][b, j] = np.maximum(heads["hm_hp"][b, j], gj * f32(0.9))
                heads["hp_offset"][b, :, pj[1], pj[0]] = kp[j] - pj
            # the network only predicts the relative size, so PnP recovers the pose in units of the object height
            objs.append({"R": R, "t": t, "scale": scale / scale[1], "height": scale[1], "kps_img": uv, "ct_int": ci})
        scenes.append(objs)
    # faint background so that top-K is tie-free without creating detections
    heads["hm"] = np.maximum(heads["hm"], (rng.rand(B, 1, H, W) * 1e-3).astype(f32))
    heads["hm_hp"] = np.maximum(heads["hm_hp"], (rng.rand(B, 8, H, W) * 1e-3).astype(f32))
    return heads, scenes


def _rot(axis, angle):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * Kx + (1 - np.cos(angle)) * Kx @ Kx


def render_video(n_frames, n_obj, seed, K=K_DEMO, img=512, out=128, sigma=1.5):
    """CenterPoseTrack-shaped heads for a short synthetic video: ``n_obj`` cuboids drift slowly (a few degrees and
    centimetres per frame); every frame carries the 11 tracking heads (opts.py:394-426).  ``tracking`` /
    ``tracking_hp`` point from the current centre / vertices to the previous frame's (dataset_combined.py builds them
    as previous - current), the two uncertainty heads are log-variances (decode.py:307-308, 325-326).
    Returns a list of per-frame head dicts (hm / hm_hp post-sigmoid, float32 NCHW, batch 1)."""
    rng = np.random.RandomState(seed)
    f32 = np.float32
    H = W = out
    ratio = out / float(img)
    ys, xs = np.mgrid[0:H, 0:W]
    objs = []
    tries = 0
    while len(objs) < n_obj and tries < 500:
        tries += 1
        scale = np.array([rng.uniform(0.6, 1.4), 1.0, rng.uniform(0.6, 1.4)]) * rng.uniform(0.18, 0.28)
        R, _ = random_pose(rng)
        t = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-1.0, 0.2), rng.uniform(2.2, 3.2)])
        axis, w = rng.randn(3), rng.uniform(0.01, 0.04)
        v = np.array([rng.uniform(-0.01, 0.01), rng.uniform(-0.01, 0.01), rng.uniform(-0.02, 0.02)])
        ok = True
        cts = []
        for f in range(n_frames):
            uv = opnp.project_points(opnp.cuboid_vertices(scale), opnp.matrix_to_rodrigues(_rot(axis, w * f) @ R), t + v * f, K)
            ok = ok and uv.min() > 24 and uv.max() < img - 24
            cts.append(uv.mean(0) * ratio)
        if not ok or any(np.linalg.norm(cts[0] - o["cts"][0]) < 28 for o in objs):
            continue
        objs.append({"scale": scale, "R": R, "t": t, "axis": axis, "w": w, "v": v, "cts": cts})
    frames = []
    prev = None
    for f in range(n_frames):
        h = {"hm": np.zeros((1, 1, H, W), f32), "hm_hp": np.zeros((1, 8, H, W), f32), "hps": np.zeros((1, 16, H, W), f32),
             "wh": np.zeros((1, 2, H, W), f32), "reg": np.zeros((1, 2, H, W), f32), "hp_offset": np.zeros((1, 2, H, W), f32),
             "scale": np.ones((1, 3, H, W), f32), "hps_uncertainty": np.full((1, 16, H, W), np.log(0.8 ** 2), f32),
             "scale_uncertainty": np.full((1, 3, H, W), np.log(0.05 ** 2), f32),
             "tracking": np.zeros((1, 2, H, W), f32), "tracking_hp": np.zeros((1, 16, H, W), f32)}
        cur = []
        for i, o in enumerate(objs):
            uv = opnp.project_points(opnp.cuboid_vertices(o["scale"]), opnp.matrix_to_rodrigues(_rot(o["axis"], o["w"] * f) @ o["R"]),
                                     o["t"] + o["v"] * f, K)
            kp = uv * ratio
            x0, y0, x1, y1 = kp[:, 0].min(), kp[:, 1].min(), kp[:, 0].max(), kp[:, 1].max()
            ct = np.array([(x0 + x1) / 2, (y0 + y1) / 2])
            ci = np.floor(ct).astype(int)
            g = np.exp(-((xs - ci[0]) ** 2 + (ys - ci[1]) ** 2) / (2 * sigma ** 2)).astype(f32)
            h["hm"][0, 0] = np.maximum(h["hm"][0, 0], g * f32(0.9 - 0.1 * i))
            h["wh"][0, :, ci[1], ci[0]] = [x1 - x0, y1 - y0]
            h["reg"][0, :, ci[1], ci[0]] = ct - ci
            h["scale"][0, :, ci[1], ci[0]] = o["scale"] / o["scale"][1] * (1.0 + 0.02 * ((f + i) % 3))
            for j in range(8):
                h["hps"][0, 2 * j:2 * j + 2, ci[1], ci[0]] = kp[j] - ci + 0.3 * np.sin([f + j, f - j])  # regression jitter
                pj = np.floor(kp[j]).astype(int)
                gj = np.exp(-((xs - pj[0]) ** 2 + (ys - pj[1]) ** 2) / (2 * sigma ** 2)).astype(f32)
                h["hm_hp"][0, j] = np.maximum(h["hm_hp"][0, j], gj * f32(0.85))
                h["hp_offset"][0, :, pj[1], pj[0]] = kp[j] - pj
            if prev is not None:
                pct, pkp = prev[i]
                h["tracking"][0, :, ci[1], ci[0]] = pct - ct
                for j in range(8):
                    h["tracking_hp"][0, 2 * j:2 * j + 2, ci[1], ci[0]] = pkp[j] - kp[j]
            cur.append((ct, kp))
        prev = cur
        h["hm"] = np.maximum(h["hm"], (rng.rand(1, 1, H, W) * 1e-3).astype(f32))
        h["hm_hp"] = np.maximum(h["hm_hp"], (rng.rand(1, 8, H, W) * 1e-3).astype(f32))
        frames.append(h)
    return frames
