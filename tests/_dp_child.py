"""child script of tests/test_dp_gloo.py::test_launch_ranks_*: one rank of a gloo group started by dp.launch_ranks"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from snn_automotive_object_detection_amd import dp

mode = sys.argv[1]
rank, local, world = dp.init_distributed(backend="gloo", timeout_s=30)
assert world == int(os.environ["WORLD_SIZE"]) and rank == int(os.environ["RANK"]) == local
if mode == "die" and rank == 1:
    sys.exit(7)                                            # a rank that fails: the launcher must stop the others
if mode == "die":
    import time
    time.sleep(60)                                         # would outlive the test if the launcher did not stop it
payload = torch.full((2, 4, 6), float(rank))
counts = torch.tensor([rank + 1, 0], dtype=torch.int32)
g_p, g_c = dp.all_gather_detection_tensors(payload, counts)
assert g_p.shape == (2 * world, 4, 6) and g_c.tolist() == [c for r in range(world) for c in (r + 1, 0)]
assert all(float(g_p[2 * r].min()) == float(r) for r in range(world))
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("LAUNCH_OK world=%d" % world, flush=True)
