"""ORACLE — test infrastructure only. Never imported by the product path.

CPU restatement (pure torch, fp32) of the four Norse 0.0.7 symbols the reference's hot path
uses.  Norse itself is a third-party dependency that is NOT vendored under /root/reference
(pinned only in prose: reference README.md:13 "norse==0.0.7"); its published algorithm is
restated here from the upstream package layout

    norse/torch/functional/lif.py              LIFParameters, lif_feed_forward_step, lif_current_encoder
    norse/torch/functional/leaky_integrator.py LIParameters, li_feed_forward_step
    norse/torch/functional/threshold.py        threshold("super") == heaviside in the forward pass
    norse/torch/module/lif.py                  LIFCell      (SNNCell wrapper, state fallback v=v_leak, i=0)
    norse/torch/module/leaky_integrator.py     LICell

Call sites in the reference that fix how they are used:
    rpn.py:16-19, 58, 67, 71, 75, 101, 106, 111, 115
    faster_rcnn.py:24-27, 444, 449, 452, 456, 468, 494, 499, 501, 506, 510

PARITY STATUS: the *neuron arithmetic* is "parity unpinned" by the reference (it has no tests and
does not ship Norse).  It is pinned here by known-answer sequences (tests/test_oracle_kat.py):
the LIF sequence is the one of upstream Norse's test_lif_feed_forward_step.  The LI update order
is the one material uncertainty; both orders are implemented behind
``li_order`` with "jump_first" (upstream li_feed_forward_step) as default.

Every operation is written in the same order as upstream so that the fp32 rounding sequence is
identical: ``dt * tau_inv`` is a 0-dim fp32 tensor product evaluated first, then multiplied
element-wise; no fused multiply-add anywhere (torch CPU element-wise kernels do not contract).
"""
from typing import NamedTuple, Optional, Tuple

import torch


class LIFParameters(NamedTuple):
    # norse/torch/functional/lif.py: defaults of LIFParameters (0-dim fp32 tensors)
    tau_syn_inv: torch.Tensor = torch.as_tensor(1.0 / 5e-3)
    tau_mem_inv: torch.Tensor = torch.as_tensor(1.0 / 1e-2)
    v_leak: torch.Tensor = torch.as_tensor(0.0)
    v_th: torch.Tensor = torch.as_tensor(1.0)
    v_reset: torch.Tensor = torch.as_tensor(0.0)
    method: str = "super"
    alpha: float = torch.as_tensor(100.0)


class LIParameters(NamedTuple):
    # norse/torch/functional/leaky_integrator.py
    tau_syn_inv: torch.Tensor = torch.as_tensor(1.0 / 5e-3)
    tau_mem_inv: torch.Tensor = torch.as_tensor(1.0 / 1e-2)
    v_leak: torch.Tensor = torch.as_tensor(0.0)


class LIFFeedForwardState(NamedTuple):
    v: torch.Tensor
    i: torch.Tensor


class LIState(NamedTuple):
    v: torch.Tensor
    i: torch.Tensor


def heaviside(data: torch.Tensor) -> torch.Tensor:
    # norse/torch/functional/heaviside.py: torch.gt(data, 0).to(data.dtype)
    return torch.gt(data, torch.as_tensor(0.0)).to(data.dtype)


def threshold(x: torch.Tensor, method: str, alpha) -> torch.Tensor:
    # forward value of every surrogate ("super" included) is the heaviside step
    return heaviside(x)


def lif_current_encoder(input_current: torch.Tensor, voltage: torch.Tensor,
                        p: LIFParameters = LIFParameters(), dt: float = 0.001
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
    """norse.torch.functional.lif.lif_current_encoder — reference call sites rpn.py:101,
    faster_rcnn.py:494."""
    dv = dt * p.tau_mem_inv * ((p.v_leak - voltage) + input_current)
    voltage = voltage + dv
    z = threshold(voltage - p.v_th, p.method, p.alpha)
    voltage = voltage - z * (voltage - p.v_reset)
    return z, voltage


def lif_feed_forward_step(input_tensor: torch.Tensor, state: LIFFeedForwardState,
                          p: LIFParameters = LIFParameters(), dt: float = 0.001
                          ) -> Tuple[torch.Tensor, LIFFeedForwardState]:
    """norse.torch.functional.lif.lif_feed_forward_step — what LIFCell applies
    (rpn.py:106; faster_rcnn.py:499,501)."""
    # compute voltage updates (uses the OLD synaptic current)
    dv = dt * p.tau_mem_inv * ((p.v_leak - state.v) + state.i)
    v_decayed = state.v + dv
    # compute current updates
    di = -dt * p.tau_syn_inv * state.i
    i_decayed = state.i + di
    # compute new spikes
    z_new = threshold(v_decayed - p.v_th, p.method, p.alpha)
    # compute reset
    v_new = (1 - z_new) * v_decayed + z_new * p.v_reset
    # compute current jumps
    i_new = i_decayed + input_tensor
    return z_new, LIFFeedForwardState(v=v_new, i=i_new)


def li_feed_forward_step(input_tensor: torch.Tensor, state: LIState,
                         p: LIParameters = LIParameters(), dt: float = 0.001,
                         li_order: str = "jump_first") -> Tuple[torch.Tensor, LIState]:
    """norse.torch.functional.leaky_integrator.li_feed_forward_step — what LICell applies
    (rpn.py:111,115; faster_rcnn.py:506,510).  ``li_order`` — SURVEY.md §8 a5."""
    if li_order == "jump_first":
        # compute current jumps
        i_new = state.i + input_tensor
        # compute voltage updates
        dv = dt * p.tau_mem_inv * ((p.v_leak - state.v) + i_new)
        v_new = state.v + dv
        # compute current updates
        di = -dt * p.tau_syn_inv * i_new
        i_decayed = i_new + di
        return v_new, LIState(v_new, i_decayed)
    elif li_order == "voltage_first":
        dv = dt * p.tau_mem_inv * ((p.v_leak - state.v) + state.i)
        v_new = state.v + dv
        di = -dt * p.tau_syn_inv * state.i
        i_decayed = state.i + di
        i_new = i_decayed + input_tensor
        return v_new, LIState(v_new, i_new)
    raise ValueError(li_order)


class LIFCell(torch.nn.Module):
    """norse.torch.module.lif.LIFCell: cell(x, state_or_None) -> (z, state).  Holds no
    parameters or buffers (reference state_dict has only conv/linear weights)."""

    def __init__(self, p: LIFParameters = LIFParameters(), dt: float = 0.001, **kwargs):
        super().__init__()
        self.p = p
        self.dt = dt

    def initial_state(self, x: torch.Tensor) -> LIFFeedForwardState:
        # SNNCell.state_fallback: v = full_like(v_leak), i = zeros
        return LIFFeedForwardState(
            v=torch.full(x.shape, float(self.p.v_leak), device=x.device, dtype=x.dtype),
            i=torch.zeros(*x.shape, device=x.device, dtype=x.dtype))

    def forward(self, x: torch.Tensor, state: Optional[LIFFeedForwardState] = None):
        if state is None:
            state = self.initial_state(x)
        return lif_feed_forward_step(x, state, self.p, self.dt)


class LICell(torch.nn.Module):
    """norse.torch.module.leaky_integrator.LICell: cell(x, state_or_None) -> (v, state)."""

    li_order = "jump_first"   # class-level switch so the shimmed reference can be run both ways

    def __init__(self, p: LIParameters = LIParameters(), dt: float = 0.001, **kwargs):
        super().__init__()
        self.p = p
        self.dt = dt

    def initial_state(self, x: torch.Tensor) -> LIState:
        return LIState(
            v=torch.full(x.shape, float(self.p.v_leak), device=x.device, dtype=x.dtype),
            i=torch.zeros(*x.shape, device=x.device, dtype=x.dtype))

    def forward(self, x: torch.Tensor, state: Optional[LIState] = None):
        if state is None:
            state = self.initial_state(x)
        return li_feed_forward_step(x, state, self.p, self.dt, li_order=type(self).li_order)
