"""Spike-rate -> energy report as a first-class call.

The reference's ``extract_spike_rates`` (/root/reference/train.py:426-517) needs hand edits of three source
files (README.md:80-88).  Here the detector returns the rate list when the two heads' ``spike_rates`` flags are
set; this module runs that mode over batches of images and evaluates the same analytical energy model:
4.6 pJ per ANN FLOP, 0.9 pJ per SNN synaptic operation (train.py:506-507), the layers at list positions
[0,3,6,9,12] (shared-LIF of the five RPN levels) and [15,16] (fc6, fc7) (train.py:482,491), and the reference's
literal factor of 1000 RoIs on the detector FLOPs (train.py:494)."""
from collections import defaultdict
from typing import Dict, Iterable, List

import torch

E_ANN_PJ = 4.6          # train.py:506
E_SNN_PJ = 0.9          # train.py:507
RPN_POSITIONS = [0, 3, 6, 9, 12]
DET_POSITIONS = [15, 16]
RPN_LAYERS = ["LVL_0", "LVL_1", "LVL_2", "LVL_3", "pool"]
DET_LAYERS = ["FC6", "FC7"]


@torch.no_grad()
def extract_spike_rates(model, batches: Iterable[List[torch.Tensor]]) -> Dict[int, torch.Tensor]:
    """Run the detector in spike-rate mode over `batches` (lists of images); returns, per list position, the
    concatenation over all batches of the [*, 2] (rate, FLOPs) tensors, on the CPU (train.py:444-467)."""
    rpn_head, det_head = model.rpn.head, model.roi_heads.box_head_and_predictor
    old = (rpn_head.spike_rates, det_head.spike_rates, model.training)
    model.eval()
    rpn_head.spike_rates = det_head.spike_rates = True
    per_layer = defaultdict(list)
    try:
        for images in batches:
            for i, t in enumerate(model(images)):
                per_layer[i].append(t.detach().cpu())
    finally:
        rpn_head.spike_rates, det_head.spike_rates = old[0], old[1]
        model.train(old[2])
    return {k: torch.cat(v, dim=0) for k, v in per_layer.items()}


def energy_report(rates: Dict[int, torch.Tensor], timesteps_rpn: int, timesteps_detector: int) -> Dict:
    """train.py:470-515 as data: per layer the mean number of spikes over the T steps, the FLOPs figure, the ANN
    and SNN energies in joule and their ratio; plus the totals."""
    layers = []
    for name, pos in zip(RPN_LAYERS, RPN_POSITIONS):
        if pos in rates:
            v = rates[pos]
            layers.append((name, float(v[:, 0].mean() * timesteps_rpn), float(v[0, 1])))
    for name, pos in zip(DET_LAYERS, DET_POSITIONS):
        if pos in rates:
            v = rates[pos]
            layers.append((name, float(v[:, 0].mean() * timesteps_detector), float(v[0, 1]) * 1000))   # literal 1000 RoIs
    out, ann_total, snn_total = [], 0.0, 0.0
    for name, mean_spikes, flops in layers:
        ann = flops * E_ANN_PJ * 1e-12
        snn = mean_spikes * flops * E_SNN_PJ * 1e-12
        out.append({"layer": name, "mean_spikes": mean_spikes, "flops": flops, "ann_energy_j": ann,
                    "snn_energy_j": snn, "snn_over_ann": snn / ann if ann else float("nan")})
        ann_total += ann
        snn_total += snn
    return {"layers": out, "ann_energy_j": ann_total, "snn_energy_j": snn_total,
            "snn_over_ann": snn_total / ann_total if ann_total else float("nan")}
