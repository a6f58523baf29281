"""Single-node multi-GPU execution of the hot path: one process per GPU, images sharded by batch.

The chain backbone -> decode -> PnP has no cross-image dependency (eval-mode BatchNorm, per-image top-K), so
CenterPose inference needs NO data-path collective: each rank processes its contiguous shard of the global
batch and scaling is "weak" (per-GPU batch fixed).  The reference has no distributed inference at all
(its only multi-GPU code is training-time ``torch.nn.DataParallel``, models/data_parallel.py:120-129, and
per-video process sharding in the evaluator, eval_video_official.py:1999-2006); this module supplies the
batch sharding both of those imply.

The one collective is for the tracking path, where the host-side tracker of a video needs the detections of
every frame: ``allgather_detections`` gathers the fixed-size detection records ([b, K, 118] float32 per rank)
with a single ``all_gather`` (RCCL over xGMI when the backend is "nccl").  Records are ~47 KB per image, so the
exchange is latency-bound; one un-bucketed all-gather of the whole shard is the right shape for point-to-point
xGMI links (no ring pipeline to fill).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            # bind the communicator to this rank's GPU up front (no lazy device guess at the first barrier)
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        try:
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        except TypeError:  # older torch: no device_id argument
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def shard_range(n_items, rank, world):
    """Contiguous [start, end) of ``n_items`` owned by ``rank``; remainders go to the lowest ranks (the same
    rule as the reference's chunk_sizes, opts.py:358-367, without its master-GPU special case)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def allgather_detections(det, group=None):
    """det: [b, K, F] float32 on this rank (same b on every rank) -> [world * b, K, F] in rank order."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return det
    world = dist.get_world_size(group)
    det = det.contiguous()
    out = torch.empty((world * det.shape[0],) + tuple(det.shape[1:]), dtype=det.dtype, device=det.device)
    dist.all_gather_into_tensor(out, det, group=group)  # concatenation along dim 0, rank order
    return out


def allgather_detections_ragged(det, counts_hint=None, group=None):
    """Variant for uneven shards (last ranks own one image fewer): pads to the largest shard, gathers, and
    returns the concatenation without the padding."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return det
    world = dist.get_world_size(group)
    n = torch.tensor([det.shape[0]], dtype=torch.int64, device=det.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    ns = [int(x.item()) for x in ns]
    m = max(ns)
    pad = torch.zeros((m,) + tuple(det.shape[1:]), dtype=det.dtype, device=det.device)
    pad[: det.shape[0]] = det
    full = allgather_detections(pad, group).view((world, m) + tuple(det.shape[1:]))
    return torch.cat([full[r, : ns[r]] for r in range(world)], 0)


def allgather_detections_compact(det, thresh, score_index=4, group=None):
    """Tracker bookkeeping only needs the detections that can start or continue a track: keep the records whose score
    (field ``score_index`` of the 118-float record, hip.DET_FIELDS['scores']) exceeds ``thresh`` -- typically < 10 of
    the K = 100 slots per image -- tag each with its global image index, and gather the ragged lists.
    det: [b, K, F] on this rank, the same b on every rank.  Returns (records [n, F], image_index [n] int64), ordered
    by rank, then image, then slot (the decode order), identical on every rank."""
    b, K = det.shape[0], det.shape[1]
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    keep = det[..., score_index] > thresh
    img = torch.arange(b, device=det.device, dtype=torch.int64).view(b, 1).expand(b, K) + rank * b
    rec = det[keep]
    idx = img[keep]
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rec, idx
    # one ragged gather: the image index rides along as an extra float64-exact column (indices < 2^24)
    packed = torch.cat([rec, idx.to(rec.dtype).unsqueeze(1)], 1)
    out = allgather_detections_ragged(packed, group=group)
    return out[:, :-1].contiguous(), out[:, -1].to(torch.int64)
