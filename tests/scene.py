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
