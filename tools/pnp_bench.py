#!/usr/bin/env python
"""Timing of cp_pnp_solve (HIP events) on synthetic cuboid scenes: N detections of 16 noisy image points each.
  python tools/pnp_bench.py            -> one line per (N, live fraction)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centerpose_amd import hip  # noqa: E402


def rodrigues(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def scenes(N, seed=0, noise=1.5, live=1.0, drop=0.0):
    rng = np.random.RandomState(seed)
    fx = fy = 663.0287679036459
    cx, cy = 300.2775065104167, 395.00066121419275
    pts = np.full((N, 16, 2), -10000.0, np.float32)
    scale = np.ones((N, 3), np.float32)
    for i in range(N):
        if rng.rand() > live:
            continue
        s = np.array([rng.uniform(0.5, 1.5), 1.0, rng.uniform(0.5, 1.5)])
        scale[i] = s
        V = np.array([[(0.5 if v & 4 else -0.5) * s[0], (0.5 if v & 2 else -0.5) * s[1], (0.5 if v & 1 else -0.5) * s[2]]
                      for v in range(8)])
        r = rng.randn(3)
        r *= rng.uniform(0.1, 3.0) / np.linalg.norm(r)
        t = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(3.0, 8.0)])
        X = V @ rodrigues(r).T + t
        uv = np.stack([fx * X[:, 0] / X[:, 2] + cx, fy * X[:, 1] / X[:, 2] + cy], 1)
        for v in range(8):
            for h in range(2):
                if rng.rand() < drop:
                    continue
                pts[i, 2 * v + h] = uv[v] + rng.randn(2) * noise
    cam = np.tile(np.array([fx, fy, cx, cy]), (N, 1))
    return pts, scale, cam


def main():
    dev = torch.device("cuda:0")
    cases = ((1, 1.0, 0.0), (64, 1.0, 0.0), (640, 1.0, 0.0), (6400, 1.0, 0.0), (6400, 0.1, 0.0), (6400, 0.1, 0.3))
    if len(sys.argv) > 1:  # python tools/pnp_bench.py 5  -> only that case (under rocprofv3: per-kernel times of one case)
        cases = cases[int(sys.argv[1]):int(sys.argv[1]) + 1]
    for N, live, drop in cases:
        timing = os.environ.get("CENTERPOSE_HIP_LIB", "").endswith("pnpt.so")
        p, s, c = scenes(N, seed=N, live=live, drop=drop)
        p, s, c = torch.from_numpy(p).to(dev), torch.from_numpy(s).to(dev), torch.from_numpy(c).to(dev)
        for _ in range(3):
            out = hip.pnp_solve(p, s, c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = hip.pnp_solve(p, s, c)
        e1.record()
        torch.cuda.synchronize()
        st = out[:, 0].cpu().numpy()
        it = out[:, 36].cpu().numpy()
        print("N %5d live %.2f drop %.1f: %8.1f us per call; status counts %s; LM iterations mean %.1f max %d; rms mean %.3f"
              % (N, live, drop, e0.elapsed_time(e1) * 100, dict(zip(*np.unique(st, return_counts=True))),
                 it[st > 0].mean() if (st > 0).any() else 0, it.max(), float(out[:, 7][out[:, 0] > 0].mean()) if (st > 0).any() else 0))
        if timing:  # -DCP_PNP_TIMING build: shader clocks of the rare detections' phases in the spare columns of their rows
            ph = out[:, 37:40].cpu().numpy()
            for i in np.nonzero(ph[:, 0] > 0)[0]:
                print("   rare detection %d (status %d, %d valid points): init %.0f clocks (EPnP eigen-decomposition %.0f), refine + pose %.0f"
                      % (i, st[i], int(out[i, 35]), ph[i, 0], ph[i, 2], ph[i, 1]))


if __name__ == "__main__":
    main()
