"""Multi-rank path on the GPU box (one device): `python bench.py --gpus 2` started WITHOUT torchrun launches its own ranks,
shards the images, runs the HIP heads per rank and exchanges the detections (gloo stand-in for RCCL, both ranks on device 0),
and the RCCL library itself is exercised with the real payload shapes in a one-rank group."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_self_launches_two_ranks_on_one_device(gpu_device):
    env = dict(os.environ, SNN_DIST_BACKEND="gloo", SNN_DP_DEVICE="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--inputs", "randn", "--no-extra", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["exchange"]["ranks"] == 2 and out["exchange"]["rows_gathered"] == 4 and out["exchange"]["ms"] > 0
    assert out["value"] > 0 and "oversubscribed" in out["config"]


def test_bench_refuses_more_ranks_than_devices_without_the_test_knob(gpu_device):
    import torch
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "SNN_DP_DEVICE", "SNN_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "device(s) visible" in r.stderr


_RCCL_ONE_RANK = r"""
import os, sys
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as dist
from snn_automotive_object_detection_amd import dp, ops
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
cls = torch.randn((2000, 9), generator=g).to(dev); reg = torch.randn((2000, 36), generator=g).to(dev)
payload, counts = ops.det_exchange_payload(cls, reg, 2, 100)
gp, gc = dp.all_gather_detection_tensors(payload, counts, force=True)       # RCCL all_gather_into_tensor on the device
torch.cuda.synchronize()
assert dist.get_backend() == "nccl" and torch.equal(gp, payload) and torch.equal(gc, counts), "gather mismatch"
t = torch.ones(1, device=dev); dist.all_reduce(t); dist.barrier()
dist.destroy_process_group()
print("RCCL_OK", tuple(gp.shape))
"""


def test_rccl_all_gather_runs_on_the_device(gpu_device):
    """the exchange step through RCCL itself (backend "nccl"), one-rank group: same call, same tensors as at N = 8"""
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK % ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "RCCL_OK (2, 100, 6)" in r.stdout, r.stderr[-3000:]


def test_dp_e2e_gathered_detections_equal_single_process(gpu_device):
    """config[3] for real at 2 ranks x 2 images (both ranks on device 0, gloo standing in for RCCL, which refuses two ranks on
    one device): whole model per rank on its contiguous image shard, all-gather of the decoded detections; the gathered list
    against the same shards run by ONE process, image by image (train.py:285-297, 598-601; coco_eval.py:158-177)."""
    env = {"SNN_DIST_BACKEND": "gloo", "SNN_DP_DEVICE": "0"}
    r = subprocess.run([sys.executable, "-c",
                        "import sys; sys.path.insert(0, %r)\nfrom snn_automotive_object_detection_amd import dp\n"
                        "sys.exit(dp.launch_ranks(%r, [], 2, timeout_s=600, extra_env=%r))" % (ROOT, os.path.join(ROOT, "tests", "_dp_e2e_child.py"), env)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")})
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DP_E2E ")][-1]
    rep = json.loads(line[len("DP_E2E "):])
    assert rep["world"] == 2 and rep["images"] == 4 and len(rep["per_image"]) == 4
    from tests._util import record_parity
    record_parity("dp_e2e_gathered_vs_single_process", **rep)
    for row in rep["per_image"]:
        # (the child runs the stock backbone's convolutions on the host: MIOpen's choices are neither repeatable run to run nor
        # the same in two processes sharing a device - tools/probe_determinism.py; everything behind the backbone is on the GPU)
        assert row.get("single_process_repeatable", True), row
        assert row["exact"], (row, rep.get("first_divergent_stage"))
