"""child script of tests/test_dp_gloo.py::test_launch_ranks_gathers_full_eval_dicts: one rank of a gloo group started by
dp.launch_ranks.  Every rank builds the (seeded) eval dicts of its contiguous shard of n_images images - boxes / scores / labels and
the optional fields all_scores / all_boxes / proposals / objectness - gathers them with dp.all_gather_detections(extras=...) and
checks the gathered list, key by key, against all images' dicts recomputed locally (= what a single process holds)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from snn_automotive_object_detection_amd import dp

n_images, K = int(sys.argv[1]), 11
SPEC = dp.ExtrasSpec(K, rois_max=40, proposals_max=64)


def eval_dict(i):
    g = torch.Generator().manual_seed(1000 + i)
    d, r, p = (3 * i) % 9, 40 - (i % 5), 64 - 3 * (i % 7)
    out = {"boxes": torch.rand((d, 4), generator=g) * 100, "scores": torch.rand((d,), generator=g),
           "labels": torch.randint(1, K, (d,), generator=g)}
    if i % 4 != 3:                                         # some images come without the optional fields
        out.update(all_scores=torch.rand((r, K), generator=g), all_boxes=torch.rand((r, K, 4), generator=g) * 100,
                   proposals=torch.rand((p, 4), generator=g) * 100, objectness=torch.rand((p,), generator=g))
    return out


rank, local, world = dp.init_distributed(backend="gloo", timeout_s=60)
mine = dp.shard_range(n_images, rank, world)
dets = [eval_dict(i) for i in mine]
for kwargs in ({}, {"images_per_rank": (n_images + world - 1) // world}):
    got = dp.all_gather_detections(dets, max_det=16, device=torch.device("cpu"), extras=SPEC, **kwargs)
    assert len(got) == n_images, (len(got), n_images)
    for i, g_ in enumerate(got):
        e = eval_dict(i)
        assert set(g_.keys()) == set(e.keys()), (i, sorted(g_.keys()), sorted(e.keys()))
        for k in e:
            assert g_[k].dtype == e[k].dtype and torch.equal(g_[k], e[k]), (i, k)
plain = dp.all_gather_detections(dets, max_det=16, device=torch.device("cpu"))          # without extras: the three base keys only
assert len(plain) == n_images and all(set(d.keys()) == {"boxes", "scores", "labels"} for d in plain)
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("GATHER_OK world=%d images=%d" % (world, n_images), flush=True)
