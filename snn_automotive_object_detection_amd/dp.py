"""Data-parallel driver pieces: one process per GPU, images sharded contiguously by rank (what the
reference's ``DistributedSampler(shuffle=False)`` does, train.py:598-601), weights replicated, and the
path's ONE exchange step: an all-gather of the per-image detections (the reference's counterpart is
the pickled ``all_gather_object`` in coco_eval.py:158-177, disabled under NCCL at train.py:874-880).

Detections are variable-length; they are padded to ``max_det`` rows and exchanged as two fixed-size
tensors (payload + counts) in ONE ``all_gather_into_tensor`` (the counts ride as an extra payload row) — KB-scale,
latency-bound, so a single collective per batch over RCCL/xGMI (backend "nccl" on ROCm) or gloo on CPU."""
import os
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: str = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "SNN_DP_DEVICE" in os.environ:
        local = int(os.environ["SNN_DP_DEVICE"])
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # SNN_DIST_BACKEND=gloo (with SNN_DP_DEVICE=0) exercises the multi-rank path on a one-GPU box
            backend = os.environ.get("SNN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_range(n_items: int, rank: int, world: int) -> range:
    """contiguous block of items for this rank (the first n_items % world ranks get one more)"""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def pack_detections(dets: List[Dict[str, torch.Tensor]], max_det: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """list of {boxes [D,4], scores [D], labels [D]} -> payload [n, max_det, 6] fp32, counts [n] i32"""
    n = len(dets)
    payload = torch.zeros((n, max_det, 6), dtype=torch.float32, device=device)
    counts = torch.zeros((n,), dtype=torch.int32, device=device)
    for i, d in enumerate(dets):
        k = min(int(d["boxes"].shape[0]), max_det)
        if k:
            payload[i, :k, 0:4] = d["boxes"][:k]
            payload[i, :k, 4] = d["scores"][:k]
            payload[i, :k, 5] = d["labels"][:k].to(torch.float32)
        counts[i] = k
    return payload, counts


def unpack_detections(payload: torch.Tensor, counts: torch.Tensor) -> List[Dict[str, torch.Tensor]]:
    out = []
    for i in range(payload.shape[0]):
        k = int(counts[i])
        out.append({"boxes": payload[i, :k, 0:4], "scores": payload[i, :k, 4],
                    "labels": payload[i, :k, 5].to(torch.int64)})
    return out


def all_gather_detection_tensors(payload: torch.Tensor, counts: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """the exchange step proper (device tensors in, device tensors out; equal per-rank image counts): ONE collective per
    batch - the per-image counts travel as an extra payload row (exact in fp32: counts < 2^24)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return payload, counts
    world = dist.get_world_size()
    n, max_det, width = payload.shape
    buf = torch.empty((n, max_det + 1, width), dtype=payload.dtype, device=payload.device)
    buf[:, :max_det] = payload
    buf[:, max_det] = counts.to(payload.dtype)[:, None]
    gathered = torch.empty((world * n, max_det + 1, width), dtype=payload.dtype, device=payload.device)
    dist.all_gather_into_tensor(gathered, buf)
    return gathered[:, :max_det], gathered[:, max_det, 0].to(counts.dtype)


def all_gather_detections(dets: List[Dict[str, torch.Tensor]], max_det: int = 1100) -> List[Dict[str, torch.Tensor]]:
    """every rank returns the detections of ALL images, in global image order (rank-major)."""
    device = dets[0]["boxes"].device if dets else torch.device("cpu")
    payload, counts = pack_detections(dets, max_det, device)
    g_payload, g_counts = all_gather_detection_tensors(payload, counts)
    return unpack_detections(g_payload, g_counts)
