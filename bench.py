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

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline      dominant kernel: algorithmic FLOPs of its launches inside the timed region / their
                HIP-event durations vs the dense matrix peak of gfx950 (MI355X_MICROARCH.md), plus the
                figures BASELINE.json's north_star names: roofline.dcn (DCNv2 + offset convolutions
                against SURVEY 8(d)'s 95.36 MB/img and the 8 TB/s HBM peak), roofline.conv1x1 (MFMA
                rate of the 1x1 convolutions), roofline.decode (microseconds and GB/s against 0.66 MB/img)
  configs2      BASELINE configs[2] (dla_34, batch 64, backbone + decode + batched PnP) timed in the
                same run at N=1, with its own dcn / conv1x1 / decode figures
  cpu_baseline  the oracle (CPU restatement of the reference graph, oracle/) timed on this host's
                cores on a bounded sample of the same workload: 1 warm-up image excluded, median over
                the timed images; "fair" (OpenMP im2col + torch CPU convolutions, all cores) is `value`,
                "faithful" (the reference's scalar single-thread im2col, as shipped) rides beside it
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
PEAK_HBM_GBPS = 8000.0  # MI355X_MICROARCH.md, HBM3E
DCN_MB_PER_IMG = 95.36  # sum over the 16 DCNv2 layers of (Cin + 27 + Cout) * HW * 4 + weights, 512x512
DCN_GFLOP_PER_IMG = 14.19  # contraction of the 16 DCNv2 layers; + 4.65 for the conv_offset_mask convolutions
DECODE_MB_PER_IMG = 0.66  # one read of hm + hm_hp, gathers, 47 KB of records
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
    ap.add_argument("--no-configs2", action="store_true", help="skip the BASELINE configs[2] leg of the default run")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--serial-pnp", action="store_true",
                    help="full: run the PnP solve on the network's stream (A/B against the side-stream default)")
    ap.add_argument("--dbg", type=int, default=0, help="cp_set_debug flags (kernel A/B switches, tuning only)")
    return ap.parse_args()


class Pipeline(object):
    """frames -> heads -> detections [-> poses], all on device."""

    def __init__(self, workload, batch, device, seed, precision="f32", serial_pnp=False):
        self.serial_pnp = serial_pnp
        self.workload = workload
        self.arch = "dlav1_34" if workload in ("decode", "track_gru") else "hourglass" if workload == "hourglass" else "dla_34"
        self.track = workload in ("track", "track_gru")
        self.heads = synth.HEADS_TRACK if self.track else synth.HEADS_POSE
        self.batch = batch
        self.device = device
        sd = synth.make_state_dict(self.arch, self.heads, self.track)
        self.model = hip.HipModel(self.arch, self.heads, sd, tracking_task=self.track, precision=precision)
        self.extra = {}
        self._stages = {}
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
                                dtype=torch.float64, device=device).repeat(batch, 1).contiguous()  # demo.py:143-144
        # 512x512 frames, fix_res: c = (256, 256), s = 512 (base_detector.py:110-114) -> inverse affine grid -> image
        from centerpose_amd.lib.utils.image import get_affine_transform
        import numpy as np
        m = np.zeros((batch, 8))
        m[:, :6] = get_affine_transform(np.array([256.0, 256.0], np.float32), 512.0, 0, (128, 128), inv=1).reshape(-1)
        m[:, 6] = 512.0 / 128
        self.meta = torch.from_numpy(m).to(device)

    def step(self, x=None, graph=False):
        if x is None:
            x, extra = self.x, self.extra
        else:
            extra = {k: v[: x.shape[0]] for k, v in self.extra.items()}
        if self.workload in ("decode", "hourglass"):
            # backbone + sigmoid + decode in one library call (hipGraph replay when graph=True)
            return self.model.detect(x, K=100, rep_mode=1, fit_gaussian=False, balance=2.0, graph=graph)[1]
        if self.workload == "full":
            det = self.model.detect(x, K=100, rep_mode=1, fit_gaussian=False, balance=2.0, graph=graph)[1]
        elif self.track:
            z = self.model(x, sigmoid_hm=True, **extra)
            det = hip.decode_raw(z["hm"], z["hps"], z["wh"], z["hm_hp"], z["hps_uncertainty"], z["scale"],
                                 z["scale_uncertainty"], z["reg"], z["hp_offset"], z["tracking"], z["tracking_hp"],
                                 K=100, rep_mode=1, fit_gaussian=True, balance=2.0)
            # the tracker of every video needs all detections: one RCCL all-gather of the fixed-size records
            return cpd.allgather_detections(det)
        # configs[2]: post-process + soft-NMS (cp_postprocess), PnP input assembly for rep_mode 1 and the batched solve
        # (cp_pnp_from_post) -- library launches only, no torch indexing and no host synchronisation inside the step.
        # The solve (a few dozen latency-bound float64 wavefronts) is queued on hip.PoseStage's side stream and runs under
        # the next batch's network; every solve has finished before the timed region's closing device synchronisation.
        n = det.shape[0]
        if self.serial_pnp:
            post, cnt = hip.postprocess(det, self.meta[:n], 0.3, nms=True)
            return det, hip.pnp_from_post(post, cnt, self.cam[:n], rep_mode=1)
        stage = self._stages.get(n)
        if stage is None:
            stage = self._stages[n] = hip.PoseStage(n, det.shape[1], self.device, depth=2)
        post, cnt, poses, done = stage.submit(det, self.meta[:n], self.cam[:n], 0.3, nms=True, rep_mode=1)
        return det, poses


def cpu_baseline(workload, arch, budget_s=40.0, n_fair=10, n_faithful=3):
    """Oracle (CPU port of the reference graph) on a bounded sample of the same workload: the first
    image is a warm-up and is not counted, `value` is 1 / median seconds per image."""
    if workload in ("track", "track_gru", "hourglass"):
        return None
    import statistics

    from oracle import backbone as ob
    from oracle import dcn as odcn
    from oracle import decode as odec

    heads = synth.HEADS_POSE
    sd = synth.make_state_dict(arch, heads, False)
    cores = torch.get_num_threads()

    def one(i, kind):
        x = synth.frames(1, seed=1000 + i)
        t1 = time.perf_counter()
        z = ob.dlaseg_forward(sd, x, heads, arch=arch.split("_")[0], dcn_kind=kind)
        hm = torch.sigmoid(z["hm"]).numpy()
        hm_hp = torch.sigmoid(z["hm_hp"]).numpy()
        odec.object_pose_decode(hm, z["hps"].numpy(), wh=z["wh"].numpy(), obj_scale=z["scale"].numpy(),
                                reg=z["reg"].numpy(), hm_hp=hm_hp, hp_offset=z["hp_offset"].numpy(), K=100,
                                rep_mode=1)
        return time.perf_counter() - t1

    def leg(kind, n_max, n_min, budget):
        one(0, kind)  # warm-up: page-in, thread pools, im2col scratch
        ts, t0 = [], time.perf_counter()
        while len(ts) < n_max and (len(ts) < n_min or time.perf_counter() - t0 < budget):
            ts.append(one(1 + len(ts), kind))
        return ts

    fair = leg("port", n_fair, 5, budget_s)
    med = statistics.median(fair)
    out = {"value": round(1.0 / med, 4), "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": "%d images of the same workload after 1 warm-up image (oracle: %s forward with OpenMP im2col + "
                     "torch CPU convolutions on %d threads, numpy decode), median %.3f s/img, min %.3f, max %.3f" % (
                         len(fair), arch, cores, med, min(fair), max(fair))}
    # faithful: the reference's CPU path as shipped -- scalar single-thread im2col (oracle/_ref, built from the
    # reference's own source where that tree exists; otherwise this repo's C port pinned to one OpenMP thread)
    kind = "reference" if odcn.have_reference() else "port"
    if kind == "port":
        try:
            import ctypes

            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(1)
        except OSError:
            kind = None
    if kind:
        ff = leg(kind, n_faithful, 2, budget_s / 2)
        fm = statistics.median(ff)
        out["faithful"] = {"value": round(1.0 / fm, 4), "unit": "images/sec", "kind": kind,
                           "sample": "%d images after 1 warm-up, scalar single-thread deformable im2col (%s) + torch CPU "
                                     "convolutions, median %.3f s/img" % (
                                         len(ff), "the reference's own dcn_v2_im2col_cpu.cpp" if kind == "reference"
                                         else "C port, 1 OpenMP thread", fm)}
    return out


def timed_region(pipe, steps, warmup, barrier):
    """W untimed steps, then exactly `steps` timed ones between barriers; per-launch HIP events on every 4th step."""
    for _ in range(warmup):
        pipe.step()
    every = 4 if steps >= 8 else 1
    sampled = 0
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        on = i % every == 0
        pipe.model.profile(on)
        sampled += int(on)
        pipe.step()
    barrier()
    dt = time.perf_counter() - t0
    pipe.model.profile(False)
    prof = pipe.model.profile_read()
    roles = pipe.model.profile_roles()
    return dt, prof, roles, sampled


def north_star_figures(roles, sampled, batch, precision):
    """roofline.dcn / conv1x1 / decode from the per-role event times (cp_model_profile_roles)."""
    out = {}
    per = lambda r: roles[r]["ms"] / sampled if r in roles else 0.0
    t_main, t_off = per("dcn"), per("dcn_offset")
    if t_main > 0:
        t = t_main + t_off
        gbps = DCN_MB_PER_IMG * batch / t  # MB / ms = GB/s
        out["dcn"] = {"bound": "hbm", "ms_per_step": round(t, 3), "main_ms": round(t_main, 3), "offset_conv_ms": round(t_off, 3),
                      "algorithmic_mb_per_img": DCN_MB_PER_IMG, "hbm_gbps": round(gbps, 1), "peak_gbps": PEAK_HBM_GBPS,
                      "frac_hbm": round(gbps / PEAK_HBM_GBPS, 4),
                      "main_only_hbm_gbps": round(DCN_MB_PER_IMG * batch / t_main, 1),
                      "tflops": round((roles["dcn"]["flops"] + roles.get("dcn_offset", {"flops": 0})["flops"]) /
                                      sampled / (t * 1e-3) / 1e12, 1),
                      "main_only_tflops": round(roles["dcn"]["flops"] / sampled / (t_main * 1e-3) / 1e12, 1)}
    peak = PEAK_F16_MFMA_TFLOPS if precision == "f16x3" else PEAK_F32_MFMA_TFLOPS
    for key, names in (("conv1x1", ("conv1x1", "head_final")),):
        ms = sum(per(n) for n in names)
        if ms > 0:
            fl = sum(roles[n]["flops"] for n in names if n in roles) / sampled
            by = sum(roles[n]["bytes"] for n in names if n in roles) / sampled
            tf = fl / (ms * 1e-3) / 1e12
            out[key] = {"bound": "mfma", "ms_per_step": round(ms, 3), "tflops": round(tf, 1), "peak_tflops": peak,
                        "mfma_utilisation": round(tf / peak, 4),
                        "issued_utilisation": round((3 if precision == "f16x3" else 1) * tf / peak, 4),
                        "algorithmic_gbps": round(by / 1e6 / ms, 1),
                        "launches_per_step": sum(roles[n]["launches"] for n in names if n in roles) // sampled}
    if "decode" in roles:
        us = per("decode") * 1e3
        out["decode"] = {"bound": "hbm", "us_per_step": round(us, 1), "algorithmic_mb_per_img": DECODE_MB_PER_IMG,
                         "hbm_gbps": round(DECODE_MB_PER_IMG * batch / (us * 1e-3), 1),
                         "floor_us_at_hbm_peak": round(DECODE_MB_PER_IMG * batch / PEAK_HBM_GBPS * 1e3, 2)}
    out["ms_per_step_by_role"] = {r: round(v["ms"] / sampled, 3) for r, v in roles.items()}
    return out


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
    if args.dbg:
        hip.lib().cp_set_debug(args.dbg)
    batch = args.batch or {"decode": 32, "full": 64, "track": 16, "track_gru": 16, "hourglass": 8}[args.workload]
    pipe = Pipeline(args.workload, batch, device, seed=317 + 1000 * rank, precision=args.precision,
                    serial_pnp=args.serial_pnp)

    def barrier():
        if dist is not None:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    side = torch.cuda.Stream(device=device)  # a non-default stream (hipGraph capture needs one)
    torch.cuda.set_stream(side)
    dt, prof, roles, sampled = timed_region(pipe, args.steps, args.warmup, barrier)
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
            roof.update(north_star_figures(roles, sampled, batch, args.precision))
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
        cfg2 = None
        if world == 1 and args.workload == "decode" and not args.no_configs2:
            # BASELINE configs[2] in the same run: dla_34, batch 64, backbone + decode + batched PnP
            del pipe.model
            torch.cuda.empty_cache()
            p2 = Pipeline("full", 64, device, seed=317, precision=args.precision, serial_pnp=args.serial_pnp)
            k2 = max(4, min(args.steps, 12))
            dt2, prof2, roles2, sampled2 = timed_region(p2, k2, max(1, min(args.warmup, 2)), barrier)
            n2, r2 = max(prof2.items(), key=lambda kv: kv[1]["ms"])
            cfg2 = {"workload": "dla_34 512x512 batch=64, Objectron-shaped synthetic frames, backbone + sigmoid + decode + "
                                "batched PnP (BASELINE configs[2])",
                    "pnp": ("on the network's stream" if args.serial_pnp else
                            "hip.PoseStage: queued on a side stream, runs under the next batch's network; all solves finish "
                            "inside the timed region"),
                    "value": round(64 * k2 / dt2, 2), "unit": "images/sec", "steps": k2, "ms_per_step": round(dt2 / k2 * 1e3, 3),
                    "whole_step_tflops": round(64 * k2 / dt2 * GFLOP_PER_IMG["dla_34"] / 1e3, 2),
                    "dominant_kernel": {"kernel": n2, "tflops": round(r2["flops"] / (r2["ms"] * 1e-3) / 1e12, 1),
                                        "avg_launch_us": round(r2["ms"] * 1e3 / r2["launches"], 2)}}
            cfg2.update(north_star_figures(roles2, sampled2, 64, args.precision))
            if not args.no_latency:  # per-frame latency of the whole chain at batch 1: network graph + post-process + PnP
                x1 = p2.x[:1].contiguous()
                for _ in range(3):
                    p2.step(x1, graph=True)
                torch.cuda.synchronize()
                ts = []
                for _ in range(50):
                    t1 = time.perf_counter()
                    p2.step(x1, graph=True)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t1) * 1e3)
                ts.sort()
                cfg2["p50_frame_ms_batch1"] = round(ts[len(ts) // 2], 3)
            del p2
            torch.cuda.empty_cache()
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
                       "gflop_per_image": gf,
                       **({"gflop_note": "603.7 = BASELINE.md's 739.2 GFLOP/img for the reference module minus the 135.3 of the "
                                         "first stack's seven heads, which do not feed model(x)[-1] (object_pose.py:135) and "
                                         "are not computed"} if pipe.arch == "hourglass" else {})},
            "p50_frame_ms_batch1": lat,
            "whole_step_tflops": round(value * gf / 1e3 / world, 2),
            "roofline": roof, "configs2": cfg2, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        barrier()  # rank 0 may still be measuring the batch-1 latency; leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
