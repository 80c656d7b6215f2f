"""Data-parallel driver pieces: one process per GPU, images sharded contiguously by rank (what the
reference's ``DistributedSampler(shuffle=False)`` does, train.py:598-601), weights replicated, and the
path's ONE exchange step: an all-gather of the per-image detections (the reference's counterpart is
the pickled ``all_gather_object`` in coco_eval.py:158-177, disabled under NCCL at train.py:874-880).
Process-group set-up follows the reference's env-driven ``init_distributed_mode`` (utils.py:268-312:
RANK / WORLD_SIZE / LOCAL_RANK, backend "nccl" = RCCL on ROCm) with an explicit collective time-out.

Detections are variable-length; they are padded to ``max_det`` rows and exchanged as two fixed-size
tensors (payload + counts) in ONE ``all_gather_into_tensor`` (the counts ride as an extra payload row) — KB-scale,
latency-bound, so a single collective per batch over RCCL/xGMI (backend "nccl" on ROCm) or gloo on CPU.
Ranks may hold different numbers of images (``shard_range`` hands the first ``n % world`` ranks one more):
``all_gather_detections`` pads every rank's block to the largest one with count -1 rows and drops them after the gather."""
import datetime
import os
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

DEFAULT_TIMEOUT_S = 300.0           # collective time-out (SNN_DIST_TIMEOUT_S overrides): ranks of a fresh node differ by the
                                    # seconds MIOpen / hipcc first-use work takes, never by minutes


def dist_timeout_s() -> float:
    return float(os.environ.get("SNN_DIST_TIMEOUT_S", DEFAULT_TIMEOUT_S))


def init_distributed(backend: str = None, timeout_s: float = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; no-op for a single process.
    ``timeout_s``: every collective (and the rendezvous) fails after this many seconds instead of hanging."""
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
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")    # a timed-out collective aborts the rank
        timeout = datetime.timedelta(seconds=dist_timeout_s() if timeout_s is None else timeout_s)
        with _stdout_to_stderr():        # gloo announces "[Gloo] Rank 0 is connected to ..." on stdout: rank 0's stdout is the JSON line's
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, local, world


class _stdout_to_stderr:
    """file descriptor 1 -> 2 for the duration (native libraries write to the descriptor, not to sys.stdout)"""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def backend_name() -> str:
    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else "none"


def shard_range(n_items: int, rank: int, world: int) -> range:
    """contiguous block of items for this rank (the first n_items % world ranks get one more)"""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


class ExtrasSpec:
    """caps of the OPTIONAL per-image fields of the reference's eval dicts: ``all_scores [R, K]``, ``all_boxes [R, K, 4]``
    (roi_heads.py:1247-1255: what new_object_discovery.py reads) and the RPN's pre-NMS report ``proposals [P, 4]``, ``objectness [P]``
    (rpn.py:493-499, generalized_rcnn.py:125-132).  Every rank must use the same spec (shapes of a collective).  Defaults: the
    reference's test-time limits - 1000 RoIs per image (model.py:51-53), 4 x 1000 + 864 pre-NMS candidates."""

    def __init__(self, num_classes: int, rois_max: int = 1000, proposals_max: int = 4864):
        self.K, self.R, self.P = int(num_classes), int(rois_max), int(proposals_max)

    HEADER = 4                                     # R, K, P of the image + a presence mask

    @property
    def width(self) -> int:
        return self.HEADER + self.R * self.K * 5 + self.P * 5


def _pack_extras(dets: List[Dict[str, torch.Tensor]], spec: "ExtrasSpec", device) -> torch.Tensor:
    """[n, spec.width] fp32: header (R, K, P, presence bits), all_scores, all_boxes, proposals, objectness - each padded to its cap"""
    n = len(dets)
    out = torch.zeros((n, spec.width), dtype=torch.float32, device=device)
    o_s, o_b = spec.HEADER, spec.HEADER + spec.R * spec.K
    o_p = o_b + spec.R * spec.K * 4
    o_o = o_p + spec.P * 4
    for i, d in enumerate(dets):
        present = 0
        r = k = pn = 0
        if "all_scores" in d and "all_boxes" in d:
            sc, bx = d["all_scores"], d["all_boxes"]
            r, k = int(sc.shape[0]), int(sc.shape[1]) if sc.dim() == 2 else 0
            if r > spec.R or (r and k != spec.K) or tuple(bx.shape) != (r, k, 4):
                raise ValueError("image %d: all_scores %s / all_boxes %s do not fit ExtrasSpec(num_classes=%d, rois_max=%d)"
                                 % (i, tuple(sc.shape), tuple(bx.shape), spec.K, spec.R))
            out[i, o_s:o_s + r * k] = sc.reshape(-1)
            out[i, o_b:o_b + r * k * 4] = bx.reshape(-1)
            present |= 1
        if "proposals" in d and "objectness" in d:
            pr, ob = d["proposals"], d["objectness"]
            pn = int(pr.shape[0])
            if pn > spec.P or tuple(ob.shape) != (pn,):
                raise ValueError("image %d: %d proposals do not fit ExtrasSpec(proposals_max=%d)" % (i, pn, spec.P))
            out[i, o_p:o_p + pn * 4] = pr.reshape(-1)
            out[i, o_o:o_o + pn] = ob
            present |= 2
        out[i, 0], out[i, 1], out[i, 2], out[i, 3] = float(r), float(k), float(pn), float(present)
    return out


def _unpack_extras(extra: torch.Tensor, spec: "ExtrasSpec", row: int, into: Dict[str, torch.Tensor]) -> None:
    o_s, o_b = spec.HEADER, spec.HEADER + spec.R * spec.K
    o_p = o_b + spec.R * spec.K * 4
    o_o = o_p + spec.P * 4
    r, k, pn, present = (int(v) for v in extra[row, :4].tolist())
    if present & 1:
        into["all_scores"] = extra[row, o_s:o_s + r * k].reshape(r, k)
        into["all_boxes"] = extra[row, o_b:o_b + r * k * 4].reshape(r, k, 4)
    if present & 2:
        into["proposals"] = extra[row, o_p:o_p + pn * 4].reshape(pn, 4)
        into["objectness"] = extra[row, o_o:o_o + pn]


def pack_detections(dets: List[Dict[str, torch.Tensor]], max_det: int, device, extras: Optional["ExtrasSpec"] = None):
    """list of {boxes [D,4], scores [D], labels [D]} -> payload [n, max_det, 6] fp32, counts [n] i32
    (+ ``extras``: a third tensor [n, extras.width] with the optional fields, see ExtrasSpec)"""
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
    if extras is not None:
        return payload, counts, _pack_extras(dets, extras, device)
    return payload, counts


def unpack_detections(payload: torch.Tensor, counts: torch.Tensor, extra: Optional[torch.Tensor] = None,
                      extras: Optional["ExtrasSpec"] = None) -> List[Dict[str, torch.Tensor]]:
    """rows with a negative count are padding of a short rank and are dropped"""
    out = []
    cnt = counts.tolist()
    for i in range(payload.shape[0]):
        k = int(cnt[i])
        if k < 0:
            continue
        d = {"boxes": payload[i, :k, 0:4], "scores": payload[i, :k, 4], "labels": payload[i, :k, 5].to(torch.int64)}
        if extra is not None:
            _unpack_extras(extra, extras, i, d)
        out.append(d)
    return out


def _collective_all_gather(out: torch.Tensor, inp: torch.Tensor) -> None:
    """all_gather_into_tensor; gloo has no device collectives for this op, so device tensors are staged through the
    host there (test configuration only: N ranks on one GPU)"""
    if inp.is_cuda and dist.get_backend() == "gloo":
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, inp.cpu())
        out.copy_(h_out)
    else:
        dist.all_gather_into_tensor(out, inp)


def all_gather_detection_tensors(payload: torch.Tensor, counts: torch.Tensor, force: bool = False,
                                 extra: Optional[torch.Tensor] = None):
    """the exchange step proper (device tensors in, device tensors out): ONE collective per batch - the per-image
    counts travel as an extra payload row (exact in fp32: |counts| < 2^24), and the optional fields (``extra`` [n, E], see
    ExtrasSpec) behind it in the same buffer.  Every rank must pass the same number of images n (all_gather_into_tensor needs
    equal shapes; ``all_gather_detections`` pads short ranks).  ``force``: run the collective even in a one-rank group
    (exercises RCCL on a single GPU).  Returns (payload, counts) or, with ``extra``, (payload, counts, extra)."""
    active = dist.is_available() and dist.is_initialized()
    if not active or (dist.get_world_size() == 1 and not force):
        return (payload, counts) if extra is None else (payload, counts, extra)
    world = dist.get_world_size()
    n, max_det, width = payload.shape
    base = (max_det + 1) * width
    e_w = 0 if extra is None else int(extra.shape[1])
    buf = torch.empty((n, base + e_w), dtype=payload.dtype, device=payload.device)
    rows = buf[:, :base].view(n, max_det + 1, width)
    rows[:, :max_det] = payload
    rows[:, max_det] = counts.to(payload.dtype)[:, None]
    if extra is not None:
        buf[:, base:] = extra
    gathered = torch.empty((world * n, base + e_w), dtype=payload.dtype, device=payload.device)
    _collective_all_gather(gathered, buf)
    g_rows = gathered[:, :base].view(world * n, max_det + 1, width)
    out = (g_rows[:, :max_det], g_rows[:, max_det, 0].to(counts.dtype))
    return out if extra is None else out + (gathered[:, base:],)


def all_gather_detections(dets: List[Dict[str, torch.Tensor]], max_det: int = 1100,
                          device: Optional[torch.device] = None, images_per_rank: Optional[int] = None,
                          extras: Optional["ExtrasSpec"] = None) -> List[Dict[str, torch.Tensor]]:
    """every rank returns the detections of ALL images, in global image order (rank-major).  Ranks may pass
    different numbers of images (also zero): the block size is agreed with one MAX all-reduce first - unless the caller
    states it (``images_per_rank`` = the largest image count of any rank, e.g. ceil(batch / world) for ``shard_range``):
    then the all-gather is the ONLY collective of the batch.
    ``extras`` (an ExtrasSpec, the same on every rank): the gathered dicts also carry ``all_scores`` / ``all_boxes`` /
    ``proposals`` / ``objectness`` of every image that had them - the reference's full eval dict (roi_heads.py:1247-1255,
    generalized_rcnn.py:125-132), still in ONE collective.  Without it those fields are NOT exchanged (boxes / scores / labels only)."""
    if device is None:
        device = dets[0]["boxes"].device if dets else torch.device("cpu")
    packed = pack_detections(dets, max_det, device, extras)
    payload, counts = packed[0], packed[1]
    extra = packed[2] if extras is not None else None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if images_per_rank is not None:
            if images_per_rank < len(dets):
                raise ValueError("images_per_rank = %d but this rank holds %d images" % (images_per_rank, len(dets)))
            n_max = int(images_per_rank)
        else:
            n_max = torch.tensor([len(dets)], dtype=torch.int64, device=device if dist.get_backend() != "gloo" else "cpu")
            dist.all_reduce(n_max, op=dist.ReduceOp.MAX)
            n_max = int(n_max.item())
        if n_max > len(dets):                              # short rank: pad with rows marked count = -1
            pad = n_max - len(dets)
            payload = torch.cat([payload, payload.new_zeros((pad, max_det, 6))], 0)
            counts = torch.cat([counts, counts.new_full((pad,), -1)], 0)
            if extra is not None:
                extra = torch.cat([extra, extra.new_zeros((pad, extra.shape[1]))], 0)
    if extra is None:
        g_payload, g_counts = all_gather_detection_tensors(payload, counts)
        return unpack_detections(g_payload, g_counts)
    g_payload, g_counts, g_extra = all_gather_detection_tensors(payload, counts, extra=extra)
    return unpack_detections(g_payload, g_counts, g_extra, extras)


# ---------------------------------------------------------------------------------------------
# rank launcher: `python bench.py --gpus N` without torchrun (the parent never touches the GPU)
# ---------------------------------------------------------------------------------------------
def count_gpus_without_hip() -> Optional[int]:
    """GPUs visible to this job, counted WITHOUT opening HIP in the calling process: a short-lived child asks torch
    (``torch.cuda.device_count()``) and exits before any rank is started.  The launcher parent must stay GPU-less - on this
    pool a process that has initialised the GPU must not start or become another GPU program - and the kernel driver's
    sysfs topology is no substitute: inside a container it lists every GPU of the node, not the ones this job may open.
    None if the child fails (the caller then trusts --gpus)."""
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print('SNN_NGPU', torch.cuda.device_count())"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
        for line in r.stdout.splitlines():
            if line.startswith("SNN_NGPU "):
                return int(line.split()[1])
    except (OSError, ValueError, subprocess.TimeoutExpired):
        pass
    return None


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(script: str, argv: Sequence[str], world: int, timeout_s: float = 1500.0, extra_env: Dict[str, str] = None) -> int:
    """Start ``world`` child processes of ``script`` (one per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment, utils.py:268-275's contract), pass rank 0's stdout through, and return 0 only if every rank exits 0.
    A rank that dies or the time-out takes the others down (terminate, then kill).  The caller must not have
    initialised the GPU: children are started with subprocess (fork + exec of a GPU-less parent)."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if extra_env:
            env.update(extra_env)
        out = None if r == 0 else subprocess.DEVNULL         # rank 0 prints the JSON line; errors of all ranks reach stderr
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=out))
    deadline = time.time() + timeout_s
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is not None:
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print("rank %d exited with code %d; stopping the other ranks" % (procs.index(p), code), file=sys.stderr)
        if alive and (rc != 0 or time.time() > deadline):
            if rc == 0:
                rc = 124
                print("ranks still running after %.0f s; stopping them" % timeout_s, file=sys.stderr)
            for p in alive:
                p.terminate()
            t_end = time.time() + 10
            for p in alive:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    return rc
