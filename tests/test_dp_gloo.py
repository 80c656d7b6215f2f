"""Multi-process (world_size 2, gloo, CPU) test of the data-parallel pieces: contiguous image sharding
and the all-gather of padded per-image detections (the path's only exchange step)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make_dets(img_index: int):
    g = torch.Generator().manual_seed(100 + img_index)
    d = 3 + 2 * img_index                                 # ragged: 3, 5, 7, 9 detections
    if img_index == 2:
        d = 0                                             # an image without detections
    return {"boxes": torch.rand((d, 4), generator=g) * 100, "scores": torch.rand((d,), generator=g),
            "labels": torch.randint(1, 9, (d,), generator=g)}


def _worker(rank, world, port, n_images, q, max_det=8):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from snn_automotive_object_detection_amd import dp
    r, l, w = dp.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    mine = dp.shard_range(n_images, rank, world)
    dets = [_make_dets(i) for i in mine]
    gathered = dp.all_gather_detections(dets, max_det=8)
    # the caller may state the largest shard itself: then the all-gather is the only collective (same result)
    stated = dp.all_gather_detections(dets, max_det=8, images_per_rank=(n_images + world - 1) // world)
    ok = len(gathered) == n_images == len(stated)
    for a, b in zip(gathered, stated):
        ok &= all(torch.equal(a[k], b[k]) for k in ("boxes", "scores", "labels"))
    for i, d in enumerate(gathered):                      # global image order, truncated to max_det rows
        e = _make_dets(i)
        k = min(e["boxes"].shape[0], 8)
        ok &= d["boxes"].shape == (k, 4) and torch.equal(d["boxes"], e["boxes"][:k])
        ok &= torch.equal(d["scores"], e["scores"][:k]) and torch.equal(d["labels"], e["labels"][:k])
        ok &= d["labels"].dtype == torch.int64
    q.put((rank, bool(ok), list(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_detections_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 4, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == (0, True, [0, 1]) and res[1] == (1, True, [2, 3])


@pytest.mark.parametrize("n_images", [3, 1])
def test_all_gather_detections_unequal_shards_world2_gloo(n_images):
    """n_images % world != 0 (and a rank without any image): short ranks pad their block, the padding is dropped after the
    gather, every rank sees all images in global order (all_gather_into_tensor needs equal shapes)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] + res[1][2] == list(range(n_images))


def test_launch_ranks_runs_world2_and_reports_failures():
    """dp.launch_ranks (what `python bench.py --gpus N` uses without torchrun): env contract, rank 0's stdout passes
    through, and a dying rank takes the group down with a non-zero exit code instead of hanging"""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = os.path.join(root, "tests", "_dp_child.py")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from snn_automotive_object_detection_amd import dp\n"
            "sys.exit(dp.launch_ranks(%r, [sys.argv[1]], 2, timeout_s=90))\n" % (root, child))
    r = subprocess.run([sys.executable, "-c", code, "ok"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=150)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "LAUNCH_OK world=2" in r.stdout
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code, "die"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=150)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 50                           # rank 0 (sleeping 60 s) was stopped, not waited for
    assert "rank 1 exited with code 7" in r.stderr


def test_shard_range_partitions_everything():
    from snn_automotive_object_detection_amd import dp
    for n in (0, 1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            parts = [list(dp.shard_range(n, r, world)) for r in range(world)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_single_process_gather_is_identity():
    from snn_automotive_object_detection_amd import dp
    dets = [_make_dets(0), _make_dets(2)]
    out = dp.all_gather_detections(dets, max_det=16)
    assert len(out) == 2 and torch.equal(out[0]["boxes"], dets[0]["boxes"]) and out[1]["boxes"].shape == (0, 4)


@pytest.mark.parametrize("world,n_images", [(8, 32), (3, 7), (8, 5)])
def test_launch_ranks_gathers_full_eval_dicts(world, n_images):
    """pre-flight of the driver's N = 8 run on CPU (gloo): dp.launch_ranks starts `world` ranks, every rank gathers the reference's
    FULL eval dicts (boxes / scores / labels + all_scores / all_boxes / proposals / objectness: roi_heads.py:1247-1255,
    generalized_rcnn.py:125-132) of all images in one collective and finds them equal, key by key, to the single-process ones - for
    even shards (8 x 4: config[3]), uneven ones (3 ranks, 7 images) and ranks without any image (8 ranks, 5 images)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = os.path.join(root, "tests", "_dp_gather_child.py")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from snn_automotive_object_detection_amd import dp\n"
            "sys.exit(dp.launch_ranks(%r, [sys.argv[1]], %d, timeout_s=150))\n" % (root, child, world))
    r = subprocess.run([sys.executable, "-c", code, str(n_images)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "GATHER_OK world=%d images=%d" % (world, n_images) in r.stdout


def test_extras_refuse_what_does_not_fit():
    from snn_automotive_object_detection_amd import dp
    spec = dp.ExtrasSpec(5, rois_max=4, proposals_max=4)
    d = {"boxes": torch.zeros(0, 4), "scores": torch.zeros(0), "labels": torch.zeros(0, dtype=torch.int64),
         "all_scores": torch.zeros(5, 5), "all_boxes": torch.zeros(5, 5, 4)}
    with pytest.raises(ValueError, match="rois_max"):
        dp.pack_detections([d], 8, torch.device("cpu"), spec)
