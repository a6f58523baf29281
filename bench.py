#!/usr/bin/env python
"""bench.py — throughput of the CenterPose inference hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): dlav1_34 at
512x512, batch 32 per GPU, synthetic random frames, seeded random-init weights of that architecture
-> backbone forward (DLA-34 + DCNv2 up-sampling + ConvGRU + GroupNorm heads) -> sigmoid ->
heat-map decode (NMS, top-100, gathers, keypoint association) on device.  `--workload full` runs
configs[2] instead (dla_34, batch 64, backbone + decode + batched PnP).
One "step" = one batch through that chain, inputs resident in HBM before the timed region.
Images shard by batch across ranks (weak scaling, no data-path collective: the chain is per-image).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel (the f32-MFMA implicit-GEMM convolution): algorithmic FLOPs of its
                launches inside the timed region / their HIP-event durations, vs the 157.3 TFLOP/s
                dense f32 matrix peak of gfx950 (MI355X_MICROARCH.md)
  cpu_baseline  the oracle (CPU restatement of the reference graph, oracle/) timed on this host's
                cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from centerpose_amd import distributed as cpd  # noqa: E402
from centerpose_amd import hip, synth  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, dense f32 matrix
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md, dense f16/bf16 matrix
GFLOP_PER_IMG = {"dlav1_34": 106.85, "dla_34": 85.11, "dla_34_track": 109.68,
                 "dlav1_34_track": 138.7, "hourglass": 603.7}  # BASELINE.md section 2, SURVEY 8(a) M9 / 8(f) N4
# hourglass: 739.0 GFLOP/img for the reference module minus the 135.3 of the first stack's seven heads, which do not feed
# model(x)[-1] and are not computed


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="decode", choices=["decode", "full", "track", "track_gru", "hourglass"],
                    help="decode: configs[1] (default); full: configs[2] dla_34 + PnP; track: dla_34 two-frame "
                         "CenterPoseTrack inputs + Gaussian-moment decode + RCCL all-gather of detection records; "
                         "track_gru: the same on dlav1_34 (two-frame input + ConvGRU heads = BASELINE configs[4] as "
                         "the reference can actually run it, SURVEY 8(f) N4 option ii); hourglass: the 2-stack hourglass "
                         "backbone, single frame, + decode (configs[4] as the reference DEFINES it, N4 option i)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 32 / 64)")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3"],
                    help="f32: exact float32 MFMA; f16x3: split-binary16 MFMA (float32-class accuracy)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    return ap.parse_args()


class Pipeline(object):
    """frames -> heads -> detections [-> poses], all on device."""

    def __init__(self, workload, batch, device, seed, precision="f32"):
        self.workload = workload
        self.arch = "dlav1_34" if workload in ("decode", "track_gru") else "hourglass" if workload == "hourglass" else "dla_34"
        self.track = workload in ("track", "track_gru")
        self.heads = synth.HEADS_TRACK if self.track else synth.HEADS_POSE
        self.batch = batch
        self.device = device
        sd = synth.make_state_dict(self.arch, self.heads, self.track)
        self.model = hip.HipModel(self.arch, self.heads, sd, tracking_task=self.track, precision=precision)
        self.extra = {}
        if self.track:  # previous frame + rendered previous heat-maps (base_detector.py:150-388)
            g = synth._gen(seed, "pre")
            self.extra = dict(
                pre_img=torch.cat([synth.frames(min(8, batch - i), seed=seed + 500 + i) for i in range(0, batch, 8)]).to(device),
                pre_hm=(torch.rand(batch, 1, 512, 512, generator=g) ** 16).to(device),
                pre_hm_hp=(torch.rand(batch, 8, 512, 512, generator=g) ** 16).to(device))
        # distinct frames per batch slot (generated in chunks of 8 to bound host memory)
        xs = [synth.frames(min(8, batch - i), seed=seed + i).to(device) for i in range(0, batch, 8)]
        self.x = torch.cat(xs, 0).contiguous()
        self.cam = torch.tensor([663.0287679036459, 663.0287679036459, 300.2775065104167, 395.00066121419275],
                                dtype=torch.float64, device=device)  # demo.py:143-144

    def step(self, x=None, graph=False):
        if x is None:
            x, extra = self.x, self.extra
        else:
            extra = {k: v[: x.shape[0]] for k, v in self.extra.items()}
        if self.workload in ("decode", "hourglass"):
            # backbone + sigmoid + decode in one library call (hipGraph replay when graph=True)
            return self.model.detect(x, K=100, rep_mode=1, fit_gaussian=False, balance=2.0, graph=graph)[1]
        z = self.model(x, sigmoid_hm=True, **extra)
        if self.track:
            det = hip.decode_raw(z["hm"], z["hps"], z["wh"], z["hm_hp"], z["hps_uncertainty"], z["scale"],
                                 z["scale_uncertainty"], z["reg"], z["hp_offset"], z["tracking"], z["tracking_hp"],
                                 K=100, rep_mode=1, fit_gaussian=True, balance=2.0)
            # the tracker of every video needs all detections: one RCCL all-gather of the fixed-size records
            return cpd.allgather_detections(det)
        det = hip.decode_raw(z["hm"], z["hps"], z["wh"], z["hm_hp"], None, z["scale"], None, z["reg"],
                             z["hp_offset"], None, None, K=100, rep_mode=1, fit_gaussian=False, balance=2.0)
        if self.workload != "full":
            return det
        # PnP input assembly for rep_mode 1 (base_detector.py:558-566): per vertex (displacement, heat-map),
        # output-grid -> input-image scale (x4); every detection above vis_thresh 0.3 (opts.py:68)
        B, K = det.shape[0], det.shape[1]
        keep = det[..., 4] > 0.3
        d = det[keep]
        disp = d[:, 46:62].reshape(-1, 8, 1, 2)
        hmk = d[:, 78:94].reshape(-1, 8, 1, 2)
        pts = torch.cat([disp, hmk], 2).reshape(-1, 16, 2)
        pts = torch.where(pts == -10000, pts, pts * 4.0)
        poses = hip.pnp_solve(pts, d[:, 22:25], self.cam.expand(pts.shape[0], 4))
        return det, poses


def cpu_baseline(workload, arch, budget_s=12.0, max_imgs=16):
    """Oracle (CPU port of the reference graph) on a bounded sample of the same workload."""
    if workload in ("track", "track_gru", "hourglass"):
        return None
    from oracle import backbone as ob
    from oracle import decode as odec

    heads = synth.HEADS_POSE
    sd = synth.make_state_dict(arch, heads, False)
    cores = torch.get_num_threads()
    n, t0 = 0, time.time()
    while n < max_imgs and (time.time() - t0 < budget_s or n < 2):
        x = synth.frames(1, seed=1000 + n)
        z = ob.dlaseg_forward(sd, x, heads, arch=arch.split("_")[0])
        hm = torch.sigmoid(z["hm"]).numpy()
        hm_hp = torch.sigmoid(z["hm_hp"]).numpy()
        odec.object_pose_decode(hm, z["hps"].numpy(), wh=z["wh"].numpy(), obj_scale=z["scale"].numpy(),
                                reg=z["reg"].numpy(), hm_hp=hm_hp, hp_offset=z["hp_offset"].numpy(), K=100,
                                rep_mode=1)
        n += 1
    dt = time.time() - t0
    return {"value": round(n / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d images of the same workload (oracle: %s forward with OpenMP im2col + torch CPU convs, "
                      "numpy decode), %.1f s" % (n, arch, dt)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    # one GPU per rank; ($CP_BENCH_BACKEND=gloo with fewer GPUs than ranks is a plumbing rehearsal on a 1-GPU box)
    backend = os.environ.get("CP_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if world > ndev and backend == "nccl":
        raise SystemExit("bench.py: %d ranks but %d visible GPUs (RCCL needs one GPU per rank)" % (world, ndev))
    local = local % ndev
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        cpd.init_from_env(backend)
    batch = args.batch or {"decode": 32, "full": 64, "track": 16, "track_gru": 16, "hourglass": 8}[args.workload]
    pipe = Pipeline(args.workload, batch, device, seed=317 + 1000 * rank, precision=args.precision)

    def barrier():
        if dist is not None:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    side = torch.cuda.Stream(device=device)  # a non-default stream (hipGraph capture needs one)
    torch.cuda.set_stream(side)
    for _ in range(args.warmup):
        pipe.step()
    # live per-launch HIP-event timing of the conv kernels (two events per launch, ~300 per step) costs ~4 % of a step,
    # so it is armed on every 4th step of the timed region only; the roofline figures are averages over those launches
    every = 4 if args.steps >= 8 else 1
    sampled = 0
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        on = i % every == 0
        pipe.model.profile(on)
        sampled += int(on)
        pipe.step()
    barrier()
    dt = time.perf_counter() - t0
    pipe.model.profile(False)
    prof = pipe.model.profile_read()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * batch * args.steps / dt

    if rank == 0:
        # ---- roofline of the dominant kernel (largest share of event time inside the timed region) ----
        roof = None
        if prof:
            name, r = max(prof.items(), key=lambda kv: kv[1]["ms"])
            achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
            total_ms = sum(v["ms"] for v in prof.values())
            is16 = "f16x3" in name
            peak = PEAK_F16_MFMA_TFLOPS if is16 else PEAK_F32_MFMA_TFLOPS
            roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
                    "note": ("algorithmic FLOPs; each is executed as 3 binary16 MFMA products (hi*hi + hi*lo + lo*hi), so "
                             "the matrix pipe does 3x this work: issued rate %.0f TFLOP/s = %.3f of the f16 peak" % (
                                 3 * achieved, 3 * achieved / peak)) if is16 else "exact float32 MFMA",
                    "launches_per_step": r["launches"] // sampled,
                    "timed_steps_sampled": sampled,
                    "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2),
                    "flops_per_launch": r["flops"] / r["launches"],
                    "algorithmic_bytes_per_launch": r["bytes"] / r["launches"],
                    "share_of_conv_time": round(r["ms"] / total_ms, 4),
                    "all_conv_kernels": {k: {"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                             "ms_per_step": round(v["ms"] / sampled, 3),
                                             "launches_per_step": v["launches"] // sampled}
                                         for k, v in prof.items()},
                    "conv_ms_per_step": round(total_ms / sampled, 3)}
            tr = os.path.join(REPO, "profiles", "pmc_traffic.json")
            if os.path.exists(tr):
                with open(tr) as f:
                    t = json.load(f).get(name)
                if t:  # HBM-side bytes per launch from the rocprofv3 PMC passes of profiles/ (FETCH_SIZE x2 + WRITE_SIZE)
                    roof["traffic"] = t["hbm_bytes_per_launch"]
                    roof["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
        lat = None
        if not args.no_latency:
            # per-frame latency at batch 1: frame already in HBM -> detections in HBM, replayed from a hipGraph
            x1 = pipe.x[:1].contiguous()
            for _ in range(3):
                pipe.step(x1, graph=True)
            torch.cuda.synchronize()
            ts = []
            for _ in range(50):
                t1 = time.perf_counter()
                pipe.step(x1, graph=True)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t1) * 1e3)
            ts.sort()
            lat = round(ts[len(ts) // 2], 3)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.workload, pipe.arch)
        gf = GFLOP_PER_IMG[pipe.arch + ("_track" if pipe.track else "")]
        out = {
            "metric": "images/sec at 512x512 %s (backbone + heat-map decode%s)" % (
                "2-stack hourglass" if pipe.arch == "hourglass" else "DLA-34",
                " + PnP" if args.workload == "full" else " + detection all-gather" if pipe.track else ""),
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "f32 via split-f16 (f16x3) MFMA, f32 accumulate",
            "data": "synthetic",
            "config": {"workload": "%s 512x512 batch=%d/GPU, synthetic random frames, seeded random-init weights, "
                                   "backbone + sigmoid + heat-map decode%s" % (
                                       pipe.arch, batch, " + batched PnP" if args.workload == "full" else
                                       " (two-frame tracking inputs, Gaussian moments) + all-gather" if pipe.track else ""),
                       "arch": pipe.arch, "global_batch": world * batch, "input": "512x512",
                       "parallelism": "batch-shard x%d (%s)" % (
                           world, "all-gather of detection records" if pipe.track else "no collective"),
                       "gflop_per_image": gf},
            "p50_frame_ms_batch1": lat,
            "whole_step_tflops": round(value * gf / 1e3 / world, 2),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        barrier()  # rank 0 may still be measuring the batch-1 latency; leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
