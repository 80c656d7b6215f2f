"""ORACLE — test infrastructure only (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
Never imported by the product path.

CPU restatement (pure torch fp32, un-fused: the same op sequence the reference issues) of

    RPNHeadSNN.forward                 /root/reference/rpn.py:84-121
    RPNHeadSNN.forward (spike rates)   /root/reference/rpn.py:126-200   (string literal in the reference)
    FastRCNNPredictorSNNFull.forward   /root/reference/faster_rcnn.py:470-516
    FastRCNNPredictorSNNFull.forward (spike rates)  /root/reference/faster_rcnn.py:520-618

The neuron arithmetic comes from oracle/norse_restated.py (Norse 0.0.7 restated; "parity unpinned"
by the reference, pinned by known-answer tests).  The loop structure is pinned against the
reference's own forward bodies executed under import shims (oracle/make_golden.py) and the
committed fixtures in tests/golden/.

``trace=True`` additionally returns every intermediate a teacher-forced parity check needs.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .norse_restated import (LIFParameters, LIFCell, LICell, lif_current_encoder)

DT = 0.001                      # rpn.py:55, faster_rcnn.py:436
V_TH_ENC = 0.25                 # rpn.py:58, faster_rcnn.py:444
V_TH_LIF = 0.1                  # rpn.py:67, faster_rcnn.py:449,452


def _cells(li_order: str):
    p_enc = LIFParameters(v_th=torch.tensor(V_TH_ENC))
    lif = LIFCell(p=LIFParameters(alpha=100, v_th=torch.tensor(V_TH_LIF)), dt=DT)

    class _LI(LICell):
        pass
    _LI.li_order = li_order
    return p_enc, lif, _LI(dt=DT), _LI(dt=DT)


def rpn_head_forward(x: Sequence[torch.Tensor], w_shared: torch.Tensor, w_cls: torch.Tensor,
                     w_bbox: torch.Tensor, num_steps: int, li_order: str = "jump_first",
                     trace: bool = False, spike_rates: bool = False, counts_out: Optional[list] = None, cur_hook=None):
    """rpn.py:84-121.  x: list of [N,C,H,W]; w_shared [C,C,3,3]; w_cls [A,C,1,1]; w_bbox [4A,C,1,1].
    Returns (logits, bbox_reg[, rates][, traces]).  ``counts_out`` (a list) receives, per level, the exact number of
    shared-LIF spikes of every image (int64 [N]): what the fp32 rate of rpn.py:172 is the rounded mean of.
    ``cur_hook(name, step, cur) -> cur`` (tests only) may replace the input current of a LIF layer: how the dead-time-step
    statement of the HIP kernels (csrc/snn_kernels.hip: lif_windows) is checked against this restatement."""
    logits, bbox_reg, traces, all_rates = [], [], [], []
    C = w_shared.shape[0]
    A = w_cls.shape[0]
    for feature in x:                                                 # rpn.py:90
        p_enc, shared_lif, lif_obj, lif_bbox = _cells(li_order)
        v = torch.zeros(*feature.shape, device=feature.device)        # rpn.py:93
        state_shared_lif = state_obj = state_bbox = None              # rpn.py:96
        tr = {"z": [], "cur": [], "spk": [], "v": [], "i": [], "mem_obj": [], "mem_bbox": []}
        l_spk, l_obj, l_bbox = [], [], []
        n_spk = torch.zeros(feature.shape[0], dtype=torch.int64)
        for step in range(num_steps):                                 # rpn.py:98
            z, v = lif_current_encoder(input_current=feature, voltage=v, p=p_enc, dt=DT)   # :101
            cur = F.conv2d(z, w_shared, None, stride=1, padding=1)    # rpn.py:105
            if cur_hook is not None:
                cur = cur_hook("shared", step, cur)
            spk_shared, state_shared_lif = shared_lif(cur, state_shared_lif)               # :106
            cur_c = F.conv2d(spk_shared, w_cls)                       # rpn.py:110
            mem_obj, state_obj = lif_obj(cur_c, state_obj)            # rpn.py:111
            cur_b = F.conv2d(spk_shared, w_bbox)                      # rpn.py:114
            mem_bbox, state_bbox = lif_bbox(cur_b, state_bbox)        # rpn.py:115
            if counts_out is not None:
                n_spk += spk_shared.flatten(start_dim=1).sum(dim=1, dtype=torch.float64).to(torch.int64)
            if trace:
                tr["z"].append(z); tr["cur"].append(cur); tr["spk"].append(spk_shared)
                tr["v"].append(state_shared_lif.v); tr["i"].append(state_shared_lif.i)
                tr["mem_obj"].append(mem_obj); tr["mem_bbox"].append(mem_bbox)
            if spike_rates:                                           # rpn.py:163-165
                l_spk.append(spk_shared.flatten(start_dim=1))
                l_obj.append(mem_obj.flatten(start_dim=1))
                l_bbox.append(mem_bbox.flatten(start_dim=1))
        if counts_out is not None:
            counts_out.append(n_spk)
        logits.append(mem_obj)                                        # rpn.py:118
        bbox_reg.append(mem_bbox)                                     # rpn.py:119
        if trace:
            traces.append({k: torch.stack(vv) for k, vv in tr.items()})
        if spike_rates:                                               # rpn.py:171-195
            n = feature.shape[0]
            H, W = mem_obj.shape[2], mem_obj.shape[3]
            # stack().sum(dim=0) exactly as rpn.py:172-174 (summation order matters in fp32)
            r_sh = (torch.stack(l_spk).sum(dim=0) / num_steps).mean(dim=1, keepdim=True)
            r_ob = (torch.stack(l_obj).sum(dim=0) / num_steps).mean(dim=1, keepdim=True)
            r_bb = (torch.stack(l_bbox).sum(dim=0) / num_steps).mean(dim=1, keepdim=True)
            fl_sh = torch.tensor([9 * (H * W) * C * C]).repeat(n, 1)
            fl_ob = torch.tensor([1 * (H * W) * C * A * 4]).repeat(n, 1)   # labels swapped in the
            fl_bb = torch.tensor([1 * (H * W) * C * A]).repeat(n, 1)       # reference: keep as is
            all_rates += [torch.hstack((r_sh, fl_sh)), torch.hstack((r_ob, fl_ob)),
                          torch.hstack((r_bb, fl_bb))]
    out = [logits, bbox_reg]
    if spike_rates:
        out.append(all_rates)
    if trace:
        out.append(traces)
    return tuple(out)


def det_head_forward(x: torch.Tensor, w6: torch.Tensor, w7: torch.Tensor, w_cls: torch.Tensor,
                     w_bbox: torch.Tensor, num_steps: int, li_order: str = "jump_first",
                     trace: bool = False, spike_rates: bool = False, only_one_bbox: bool = False,
                     counts_out: Optional[list] = None, cur_hook=None):
    """faster_rcnn.py:470-516 (spike_rates=True: 520-618, which returns ONLY the rate list).
    x [R,C,7,7] (or [R,D]); w6 [Hd,D]; w7 [Hd,Hd]; w_cls [K,Hd]; w_bbox [4K,Hd].
    ``counts_out`` (a list) receives the exact lif6 / lif7 spike totals per RoI (two int64 [R] tensors)."""
    x = x.flatten(start_dim=1)                                        # faster_rcnn.py:473
    p_enc = LIFParameters(v_th=torch.tensor(V_TH_ENC))
    _, lif6, lif_cls, lif_bbox = _cells(li_order)
    _, lif7, _, _ = _cells(li_order)
    v = torch.zeros(*x.shape, device=x.device)                        # :484
    state_lif6 = state_lif7 = state_cls = state_bbox = None           # :487
    tr = {k: [] for k in ("z", "cur6", "spk6", "cur7", "spk7", "mem_cls", "mem_bbox")}
    R = x.shape[0]
    Hd, K, K4 = w6.shape[0], w_cls.shape[0], w_bbox.shape[0]
    if spike_rates:
        c6 = torch.zeros(R, Hd); c7 = torch.zeros(R, Hd)
        cc = torch.zeros(R, K); cb = torch.zeros(R, K4)
    n6 = torch.zeros(R, dtype=torch.int64); n7 = torch.zeros(R, dtype=torch.int64)
    for step in range(num_steps):                                     # :492
        z, v = lif_current_encoder(input_current=x, voltage=v, p=p_enc, dt=DT)     # :494
        cur6 = F.linear(z, w6)                                        # :498
        if cur_hook is not None:
            cur6 = cur_hook("fc6", step, cur6)
        spk_lif6, state_lif6 = lif6(cur6, state_lif6)                 # :499
        cur7 = F.linear(spk_lif6, w7)                                 # :500
        if cur_hook is not None:
            cur7 = cur_hook("fc7", step, cur7)
        spk_lif7, state_lif7 = lif7(cur7, state_lif7)                 # :501
        mem_cls, state_cls = lif_cls(F.linear(spk_lif7, w_cls), state_cls)         # :505-506
        mem_bbox, state_bbox = lif_bbox(F.linear(spk_lif7, w_bbox), state_bbox)    # :509-510
        if trace:
            for k, t in (("z", z), ("cur6", cur6), ("spk6", spk_lif6), ("cur7", cur7),
                         ("spk7", spk_lif7), ("mem_cls", mem_cls), ("mem_bbox", mem_bbox)):
                tr[k].append(t)
        if spike_rates:                                               # :556-560
            c6 += spk_lif6; c7 += spk_lif7; cc += mem_cls; cb += mem_bbox
        if counts_out is not None:
            n6 += spk_lif6.sum(dim=1, dtype=torch.float64).to(torch.int64)
            n7 += spk_lif7.sum(dim=1, dtype=torch.float64).to(torch.int64)
    if counts_out is not None:
        counts_out += [n6, n7]
    if spike_rates:                                                   # :568-618
        D = x.shape[1]
        rates = [(c / num_steps).mean(dim=1, keepdim=True) for c in (c6, c7, cc, cb)]
        flops = [D * Hd, Hd * Hd, Hd * K, Hd * K if only_one_bbox else Hd * K * 4]
        return [torch.hstack((r, torch.tensor([f]).repeat(R, 1))) for r, f in zip(rates, flops)]
    if trace:
        return mem_cls, mem_bbox, {k: torch.stack(vv) for k, vv in tr.items()}
    return mem_cls, mem_bbox                                          # :513-516


# ---------------------------------------------------------------------------------------------
# helpers shared by the parity tests (still oracle-side)
# ---------------------------------------------------------------------------------------------
def lif_scan_from_currents(cur: torch.Tensor, v_th: float = V_TH_LIF):
    """Teacher-forced LIF: given the input currents of every step [T, ...] return the spikes
    [T, ...] plus final (v, i) and the per-step decayed voltages (for |v-θ| margin checks)."""
    lif = LIFCell(p=LIFParameters(alpha=100, v_th=torch.tensor(v_th)), dt=DT)
    state = None
    zs, vdec = [], []
    for t in range(cur.shape[0]):
        if state is None:
            state = lif.initial_state(cur[t])
        # decayed voltage before reset (what the threshold sees)
        dv = DT * lif.p.tau_mem_inv * ((lif.p.v_leak - state.v) + state.i)
        vdec.append(state.v + dv)
        z, state = lif(cur[t], state)
        zs.append(z)
    return torch.stack(zs), state, torch.stack(vdec)


def encoder_spikes(x: torch.Tensor, num_steps: int) -> torch.Tensor:
    """[T, *x.shape] spikes of the constant-current encoder (rpn.py:101 / faster_rcnn.py:494)."""
    p_enc = LIFParameters(v_th=torch.tensor(V_TH_ENC))
    v = torch.zeros(*x.shape)
    zs = []
    for _ in range(num_steps):
        z, v = lif_current_encoder(input_current=x, voltage=v, p=p_enc, dt=DT)
        zs.append(z)
    return torch.stack(zs)


def li_last_from_spikes(spk: torch.Tensor, w: torch.Tensor, li_order: str = "jump_first",
                        conv: bool = False):
    """Teacher-forced LI head: spikes [T, ...] -> (last membrane, sum of membranes over t)."""
    _, _, li, _ = _cells(li_order)
    state = None
    acc = None
    for t in range(spk.shape[0]):
        cur = F.conv2d(spk[t], w) if conv else F.linear(spk[t], w)
        mem, state = li(cur, state)
        acc = mem.clone() if acc is None else acc + mem
    return mem, acc
