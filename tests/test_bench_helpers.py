"""bench.py's input-statistics helpers (extra.density_sweep) on the CPU: the worst-case tensor really fills every occupied nibble of the structured-sparse
planes, and the hit-rate statistic counts what csrc/snn_sparse.h's secondary plane holds (a nibble with three or four spikes of ONE period)."""
import numpy as np
import torch

import bench
from oracle import norse_restated as NR
from tests._util import dense_to_planes


def _planes(x, T):
    p = NR.LIFParameters(v_th=torch.as_tensor(0.25))
    v = torch.zeros_like(x)
    zs = []
    for _ in range(T):
        z, v = NR.lif_current_encoder(x, v, p, 0.001)
        zs.append(z)
    z = torch.stack(zs)                                         # [T, N, C, H, W]
    rows = z.permute(0, 1, 3, 4, 2).reshape(T, -1, x.shape[1]).numpy()
    return dense_to_planes(rows).to(torch.int64) & 0xffffffff, z


def test_worst_case_tensor_fills_every_occupied_nibble():
    T = 8
    g = torch.Generator().manual_seed(0)
    x = bench.worst_case_tensor((1, 128, 16, 32), g, T)
    planes, z = _planes(x, T)
    hit, blocks, ones, bits = bench.planes_hit_stats(planes)
    assert hit == blocks and blocks == 5 * (16 * 32 // 16) * (128 // 64)          # planes e_3 .. e_7, 16-position blocks, 64-k steps
    assert abs(ones / bits - 0.2) < 1e-9                                          # a fifth of the channels on each of the five periods
    first = (z.cumsum(0) == 1) & (z > 0)
    period = first.float().argmax(0) + 1
    assert set(period.unique().tolist()) == {3, 4, 5, 6, 7}
    assert (period[:, 0:4] == period[:, 0:1]).all()                               # four consecutive channels share a period


def test_planes_hit_stats_counts_nibbles_with_three_spikes_of_one_period():
    T, P, Cw = 5, 32, 2
    z = np.zeros((T, P, Cw * 32), dtype=np.float32)
    # period 3 (first spike at step 2): position 0 gets THREE channels of nibble 0, position 17 gets two (no hit), position 20 four of nibble 9 (word 1)
    z[2, 0, [0, 1, 3]] = 1
    z[2, 17, [4, 5]] = 1
    z[2, 20, [36, 37, 38, 39]] = 1
    # period 1 / 2 spikes never count (those planes stay on the dense instruction)
    z[0, 5, [8, 9, 10, 11]] = 1
    z[1, 6, [12, 13, 14]] = 1
    hit, blocks, ones, bits = bench.planes_hit_stats(dense_to_planes(z).to(torch.int64) & 0xffffffff)
    assert blocks == 2 * 2 * 1                                                    # planes e_3, e_4 x two 16-position blocks x one 64-k step
    assert hit == 2 and ones == 9                                                 # blocks (e_3, positions 0-15) and (e_3, 16-31)


def test_randn_has_some_but_not_all_blocks_hit():
    planes, _ = _planes(torch.randn((1, 256, 16, 32), generator=torch.Generator().manual_seed(1)), 8)
    hit, blocks, ones, bits = bench.planes_hit_stats(planes)
    assert 0.02 < hit / blocks < 0.4 and 0.02 < ones / bits < 0.08


def test_committed_traffic_is_quoted_only_for_the_tree_it_was_collected_on(tmp_path):
    """profiles/r6_traffic.json carries the source digest of its tree; bench.py's roofline.traffic is null (with the two digests as the reason) for any other build"""
    import json
    import os
    from snn_automotive_object_detection_amd import build
    path = os.path.join(bench.ROOT, "profiles", "r6_traffic.json")
    doc = json.load(open(path))
    assert len(doc["source_digest"]) == 64 and {"cityscapes", "stress", "bdd"} <= set(doc)
    t, src, prof = bench.load_committed_traffic(path, doc["source_digest"], "cityscapes", "bf16x3")
    assert t == doc["cityscapes"]["bf16x3"]["conv"]["hbm_bytes_per_launch"] and "same source digest" in src and "fc6" in prof and "encoders" in prof
    t, src, prof = bench.load_committed_traffic(path, "0" * 64, "cityscapes", "bf16x3")
    assert t is None and prof == {} and "traffic not quoted" in src and doc["source_digest"][:12] in src
    assert bench.load_committed_traffic(str(tmp_path / "missing.json"), doc["source_digest"], "cityscapes", "bf16x3") == (None, None, {})
    # the committed file belongs to the committed tree (fails when a kernel source changes without a new evidence run: re-run tools/r6_final.sh + r6_collect.sh)
    assert doc["source_digest"] == build.source_digest(), "profiles/r6_traffic.json is stale against csrc/ + include/"
