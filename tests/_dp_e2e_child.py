"""child of tests/test_gpu_dp.py::test_dp_e2e_gathered_detections_equal_single_process: one rank of BASELINE.json config[3] at a
reduced image count (2 per rank): create_model("bdd", 11) on this rank's contiguous shard -> dp.all_gather_detections of the
decoded detections.  Rank 0 then runs every shard itself (same batches, same weights) and prints how the gathered list
compares with that single-process result, image by image, as one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import bench
import snn_automotive_object_detection_amd as S
from snn_automotive_object_detection_amd import dp

PER_RANK = int(os.environ.get("DP_E2E_PER_RANK", "2"))
# The stock backbone is the one part of the model that is not bitwise reproducible ACROSS PROCESSES on the GPU: MIOpen's
# default convolution choices are not even repeatable run to run (tools/probe_determinism.py: pyramids differ by ~2e-5
# between two passes), and with torch.backends.cudnn.deterministic the choice still depends on the process (two ranks
# sharing one device see different free workspace).  This test is about the data-parallel machinery and the HIP path, so
# the backbone's convolutions run on the host here (oneDNN, fixed thread count: the same bits in every process);
# everything behind it - transform, RPN head, proposal selection, RoIAlign + detector head, post-processing - runs on the GPU.
torch.backends.cudnn.deterministic = True
torch.set_num_threads(8)


class HostBackbone(torch.nn.Module):
    def __init__(self, backbone):
        super().__init__()
        self.backbone = backbone.cpu()

    def forward(self, x):
        from collections import OrderedDict
        return OrderedDict((k, v.to(x.device)) for k, v in self.backbone(x.cpu()).items())


class StageTrace:
    """sha1 of every stage's output (backbone pyramid, RPN head, proposals, detector result) of one model call: when the gathered
    detections ever differ from the single-process ones, the report names the first stage whose bits differ"""
    STAGES = ("backbone", "rpn.head", "rpn", "roi_heads")

    def __init__(self, model):
        self.rows = {}
        for name in self.STAGES:
            mod = model
            for part in name.split("."):
                mod = getattr(mod, part)
            mod.register_forward_hook(lambda m, i, o, name=name: self.rows.__setitem__(name, self.digest(o)))

    @classmethod
    def digest(cls, o):
        import hashlib
        h = hashlib.sha1()

        def walk(x):
            if torch.is_tensor(x):
                h.update(x.detach().cpu().contiguous().numpy().tobytes())
            elif isinstance(x, dict):
                for k in sorted(x):
                    walk(x[k])
            elif isinstance(x, (list, tuple)):
                for y in x:
                    walk(y)
        walk(o)
        return h.hexdigest()[:16]

    def take(self):
        rows, self.rows = self.rows, {}
        return rows


rank, local, world = dp.init_distributed()
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
torch.manual_seed(4321)
model = S.create_model("bdd", 11, True, True, 0, False, False, 8, 12).to(dev).eval()
model.backbone = HostBackbone(model.backbone)
n_global = PER_RANK * world
mine = dp.shard_range(n_global, rank, world)
trace = StageTrace(model)
dets = model(bench.dp_images(mine, dev))
stages = [None] * world
dist.all_gather_object(stages, trace.take())
gathered = dp.all_gather_detections(dets, max_det=1100, device=dev)
assert len(gathered) == n_global
# this rank's own block of the gathered list is its own result, bit for bit
for j, i in enumerate(mine):
    for k in ("boxes", "scores", "labels"):
        assert torch.equal(gathered[i][k], dets[j][k][:1100]), (rank, i, k)
# every rank ends up with the same list
sig = torch.tensor([float(sum(float(d["boxes"].sum()) + float(d["scores"].sum()) + float(d["labels"].sum()) for d in gathered))], dtype=torch.float64)
sigs = [torch.zeros_like(sig) for _ in range(world)]
dist.all_gather(sigs, sig if dist.get_backend() == "gloo" else sig.to(dev))
assert all(float(x) == float(sig) for x in sigs), [float(x) for x in sigs]
if rank == 0:
    report = {"world": world, "images": n_global, "backend": dist.get_backend(), "per_image": []}
    for r in range(world):                                       # single process: the same batches, one after the other
        idx = list(dp.shard_range(n_global, r, world))
        ref = model(bench.dp_images(idx, dev))
        single = trace.take()
        diverged = [k for k in StageTrace.STAGES if stages[r].get(k) != single.get(k)]
        report.setdefault("first_divergent_stage", {})[str(r)] = diverged[0] if diverged else None
        ref2 = model(bench.dp_images(idx, dev)) if r == 0 else None
        for j, i in enumerate(idx):
            g, e = gathered[i], ref[j]
            n_g, n_e = int(g["boxes"].shape[0]), int(e["boxes"].shape[0])
            row = {"image": i, "rank": r, "n_gathered": n_g, "n_single": n_e,
                   "exact": bool(n_g == n_e and all(torch.equal(g[k], e[k]) for k in ("boxes", "scores", "labels")))}
            if n_g == n_e and n_g:
                row["max_box_diff"] = float((g["boxes"] - e["boxes"]).abs().max())
                row["max_score_diff"] = float((g["scores"] - e["scores"]).abs().max())
                row["labels_equal"] = bool(torch.equal(g["labels"], e["labels"]))
            if ref2 is not None:
                row["single_process_repeatable"] = bool(all(torch.equal(ref[j][k], ref2[j][k]) for k in ("boxes", "scores", "labels")))
            report["per_image"].append(row)
    print("DP_E2E " + json.dumps(report), flush=True)
dist.barrier()
dist.destroy_process_group()
