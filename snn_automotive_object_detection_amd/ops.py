"""Thin torch-tensor wrappers over the C ABI (include/snn_hip.h).  torch is plumbing here: device
memory, the current HIP stream, nothing else.  Every function raises if the input is not a CUDA/HIP
tensor — there is no CPU or eager fallback."""
import ctypes as C
from typing import List, NamedTuple, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import snn_params, snn_rpn_level, snn_roi_level

DT = 0.001                                   # rpn.py:55 / faster_rcnn.py:436


class LIFParameters(NamedTuple):
    """Mirror of norse.torch.LIFParameters for the attributes callers read
    (``.p_enc.v_th``: custom_utils.py:321-329, train.py:477-478).  0-dim fp32 tensors."""
    tau_syn_inv: torch.Tensor = torch.as_tensor(1.0 / 5e-3)
    tau_mem_inv: torch.Tensor = torch.as_tensor(1.0 / 1e-2)
    v_leak: torch.Tensor = torch.as_tensor(0.0)
    v_th: torch.Tensor = torch.as_tensor(1.0)
    v_reset: torch.Tensor = torch.as_tensor(0.0)
    method: str = "super"
    alpha: float = torch.as_tensor(100.0)


def make_params(p_enc: LIFParameters, p_lif: LIFParameters, dt: float = DT,
                li_order: str = "jump_first", precision: str = "bf16x3") -> snn_params:
    """Form the fp32 constants exactly as Norse forms them: 0-dim fp32 tensor products
    ``dt * tau_mem_inv`` and ``-dt * tau_syn_inv`` (norse lif.py / leaky_integrator.py)."""
    # The C ABI carries ONE set of time constants / rest / reset potentials (snn_params) for the encoder, the LIF cells and
    # the LI heads, which is what the reference builds (p_enc and p_lif differ in v_th only, rpn.py:58,67; the LI cells
    # use Norse's defaults, rpn.py:71,75).  Anything else would be silently ignored by the kernels, so it is refused.
    for name in ("tau_mem_inv", "v_leak", "v_reset"):
        if float(getattr(p_enc, name)) != float(getattr(p_lif, name)):
            raise ValueError("p_enc.%s = %r differs from p_lif.%s = %r: the kernels take one value for the encoder and the "
                             "LIF cells" % (name, float(getattr(p_enc, name)), name, float(getattr(p_lif, name))))
    li_default = LIFParameters()
    for name in ("tau_mem_inv", "tau_syn_inv", "v_leak"):
        if float(getattr(p_lif, name)) != float(getattr(li_default, name)):
            raise ValueError("p_lif.%s = %r differs from the LI cells' default %r: the time-collapsed LI heads use the default "
                             "LIParameters (rpn.py:71,75 / faster_rcnn.py:456,468)" % (
                                 name, float(getattr(p_lif, name)), float(getattr(li_default, name))))
    ca = dt * p_lif.tau_mem_inv.to(torch.float32)
    cb = -dt * p_lif.tau_syn_inv.to(torch.float32)
    assert ca.dtype == torch.float32 and cb.dtype == torch.float32
    return snn_params(float(ca), float(cb), float(p_lif.v_leak), float(p_lif.v_reset),
                      float(p_enc.v_th.to(torch.float32)), float(p_lif.v_th.to(torch.float32)),
                      {"jump_first": 0, "voltage_first": 1}[li_order], _lib.PRECISIONS[precision])


def _need_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _lib.SnnHipError("%s must live on the GPU (got %s); this path has no CPU fallback" % (what, t.device))


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def cdiv(a, b):
    return (a + b - 1) // b


# ---------------------------------------------------------------------------------------------
# weight packing
# ---------------------------------------------------------------------------------------------
def pack_conv3x3(w: torch.Tensor) -> torch.Tensor:
    _need_gpu(w, "conv weight")
    lib = _lib.load()
    w = _f32c(w)
    co, ci = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (3, 3)
    out = torch.empty(lib.snn_packed_conv3x3_elems(co, ci), dtype=torch.float32, device=w.device)
    _lib.check(lib.snn_pack_conv3x3_weight(_ptr(w), co, ci, _ptr(out), _stream()), "snn_pack_conv3x3_weight")
    return out


def pack_linear(w: torch.Tensor) -> torch.Tensor:
    _need_gpu(w, "linear weight")
    lib = _lib.load()
    w = _f32c(w)
    n, k = w.shape
    out = torch.empty(lib.snn_packed_linear_elems(n, k), dtype=torch.float32, device=w.device)
    _lib.check(lib.snn_pack_linear_weight(_ptr(w), n, k, _ptr(out), _stream()), "snn_pack_linear_weight")
    return out


def pack_heads(wa: torch.Tensor, wb: torch.Tensor, check_split: bool = True) -> torch.Tensor:
    """``check_split``: the LI-head kernels of every precision but "f32_strict" split these weights into three bf16 planes on the
    fly, so the same exactness check applies (InexactWeightSplit)"""
    _need_gpu(wa, "head weight")
    lib = _lib.load()
    wa = _f32c(wa).flatten(1)
    wb = _f32c(wb).flatten(1)
    if check_split:
        check_bf16x3_split(wa, "LI head weight %s" % (tuple(wa.shape),))
        check_bf16x3_split(wb, "LI head weight %s" % (tuple(wb.shape),))
    na, k = wa.shape
    nb = wb.shape[0]
    assert wb.shape[1] == k
    out = torch.empty(lib.snn_packed_heads_elems(na, nb, k), dtype=torch.float32, device=wa.device)
    _lib.check(lib.snn_pack_heads_weight(_ptr(wa), na, _ptr(wb), nb, k, _ptr(out), _stream()), "snn_pack_heads_weight")
    return out


def bf16x3_split_status(w: torch.Tensor) -> Tuple[int, int, int]:
    """(inexact, non-finite, subnormal-plane) counts of snn_check_bf16x3_split for a weight tensor.  One host synchronisation -
    weights are packed when they change, not per forward."""
    _need_gpu(w, "weight")
    lib = _lib.load()
    w = _f32c(w)
    status = torch.empty(3, dtype=torch.int32, device=w.device)
    _lib.check(lib.snn_check_bf16x3_split(_ptr(w), w.numel(), _ptr(status), _stream()), "snn_check_bf16x3_split")
    a, b, c = status.tolist()
    return a, b, c


def check_bf16x3_split(w: torch.Tensor, what: str = "weight") -> None:
    """raise InexactWeightSplit unless every element of w is exactly the sum of its three bf16 planes"""
    inexact, nonfinite, _ = bf16x3_split_status(w)
    if inexact or nonfinite:
        raise _lib.InexactWeightSplit(what, inexact, nonfinite)


def split_problem(*weights: torch.Tensor) -> Optional[_lib.InexactWeightSplit]:
    """None if every tensor splits exactly, else the InexactWeightSplit describing the first one that does not (the modules cache
    this per weight version and fall back to "f32_strict")"""
    for w in weights:
        try:
            check_bf16x3_split(w, "weight %s" % (tuple(w.shape),))
        except _lib.InexactWeightSplit as e:
            return e
    return None


def pack_conv3x3_bf16x3(w: torch.Tensor, check_split: bool = True) -> torch.Tensor:
    _need_gpu(w, "conv weight")
    lib = _lib.load()
    w = _f32c(w)
    if check_split:
        check_bf16x3_split(w, "conv weight %s" % (tuple(w.shape),))
    co, ci = w.shape[0], w.shape[1]
    out = torch.empty(lib.snn_packed_conv3x3_bf16x3_elems(co, ci), dtype=torch.int16, device=w.device)
    _lib.check(lib.snn_pack_conv3x3_weight_bf16x3(_ptr(w), co, ci, _ptr(out), _stream()), "snn_pack_conv3x3_weight_bf16x3")
    return out


def pack_linear_bf16x3(w: torch.Tensor, check_split: bool = True, inner: int = 0) -> torch.Tensor:
    """``inner`` > 1: the reduction index is permuted to k' = s * C + c (input column c * inner + s; snn_hip.h:
    snn_pack_linear_weight_bf16x3_perm) - pass the same value as ``w6_inner`` to the head call"""
    _need_gpu(w, "linear weight")
    lib = _lib.load()
    w = _f32c(w)
    if check_split:
        check_bf16x3_split(w, "linear weight %s" % (tuple(w.shape),))
    n, k = w.shape
    out = torch.empty(lib.snn_packed_linear_bf16x3_elems(n, k), dtype=torch.int16, device=w.device)
    if inner > 1:
        _lib.check(lib.snn_pack_linear_weight_bf16x3_perm(_ptr(w), n, k, int(inner), _ptr(out), _stream()), "snn_pack_linear_weight_bf16x3_perm")
    else:
        _lib.check(lib.snn_pack_linear_weight_bf16x3(_ptr(w), n, k, _ptr(out), _stream()), "snn_pack_linear_weight_bf16x3")
    return out


def pack_linear_mx(w: torch.Tensor) -> torch.Tensor:
    """[N, K] fp32 -> block-scaled fp6 digit planes (int32 words, csrc/snn_mx.h)"""
    _need_gpu(w, "weight")
    lib = _lib.load()
    w = _f32c(w)
    n, k = w.shape
    out = torch.empty(lib.snn_packed_linear_mx_words(n, k), dtype=torch.int32, device=w.device)
    _lib.check(lib.snn_pack_linear_weight_mx(_ptr(w), n, k, _ptr(out), _stream()), "snn_pack_linear_weight_mx")
    return out


def pack_conv3x3_mx(w: torch.Tensor) -> torch.Tensor:
    """[C_out, C_in, 3, 3] fp32 -> block-scaled fp6 digit planes, reduction index k = tap * Cp128 + ci"""
    _need_gpu(w, "weight")
    lib = _lib.load()
    w = _f32c(w)
    co, ci = w.shape[0], w.shape[1]
    out = torch.empty(lib.snn_packed_conv3x3_mx_words(co, ci), dtype=torch.int32, device=w.device)
    _lib.check(lib.snn_pack_conv3x3_weight_mx(_ptr(w), co, ci, _ptr(out), _stream()), "snn_pack_conv3x3_weight_mx")
    return out


# ---------------------------------------------------------------------------------------------
# stage-level ops (used by the teacher-forced parity tests)
# ---------------------------------------------------------------------------------------------
def spike_gemm_bf16x3(a_rows: torch.Tensor, K: int, N: int, w_packed: torch.Tensor) -> torch.Tensor:
    """a_rows int32 [M, Kw] -> cur fp32 [M, Np] on the bf16 matrix cores (exact 3-way weight split)"""
    _need_gpu(a_rows, "spike rows")
    lib = _lib.load()
    M = a_rows.shape[0]
    Np = cdiv(N, 32) * 32
    cur = torch.empty((M, Np), dtype=torch.float32, device=a_rows.device)
    _lib.check(lib.snn_spike_gemm_bf16x3(_ptr(a_rows), M, K, N, _ptr(w_packed), _ptr(cur), Np, _stream()),
               "snn_spike_gemm_bf16x3")
    return cur


def spike_gemm_lif_bf16x3(a_planes: torch.Tensor, K: int, N: int, p: snn_params, w_packed: torch.Tensor) -> torch.Tensor:
    """a_planes int32 [T, R, Kw] -> LIF spike planes int32 [T, R, Nw]: linear layer + LIF over T in one launch"""
    _need_gpu(a_planes, "spike planes")
    lib = _lib.load()
    T, R, _ = a_planes.shape
    Nw = cdiv(N, 32)
    spk = torch.empty((T, R, Nw), dtype=torch.int32, device=a_planes.device)
    _lib.check(lib.snn_spike_gemm_lif_bf16x3(_ptr(a_planes), T, R, K, N, C.byref(p), _ptr(w_packed), _ptr(spk), R * Nw,
                                             _stream()), "snn_spike_gemm_lif_bf16x3")
    return spk


def spike_gemm_mx(a_rows: torch.Tensor, K: int, N: int, w_packed: torch.Tensor) -> torch.Tensor:
    """a_rows int32 [M, Kw] -> cur fp32 [M, Np] on the fp4 x fp6 block-scaled matrix path"""
    _need_gpu(a_rows, "spike rows")
    lib = _lib.load()
    M, Np = a_rows.shape[0], cdiv(N, 32) * 32
    cur = torch.empty((M, Np), dtype=torch.float32, device=a_rows.device)
    _lib.check(lib.snn_spike_gemm_mx(_ptr(a_rows), M, K, N, _ptr(w_packed), _ptr(cur), Np, _stream()), "snn_spike_gemm_mx")
    return cur


def spike_gemm_lif_mx(a_planes: torch.Tensor, K: int, N: int, p: snn_params, w_packed: torch.Tensor) -> torch.Tensor:
    """a_planes int32 [T, R, Kw] -> LIF spike planes int32 [T, R, Nw] (fp4 x fp6 path, LIF fused in the row tile)"""
    _need_gpu(a_planes, "spike planes")
    lib = _lib.load()
    T, R, _ = a_planes.shape
    Nw = cdiv(N, 32)
    spk = torch.empty((T, R, Nw), dtype=torch.int32, device=a_planes.device)
    _lib.check(lib.snn_spike_gemm_lif_mx(_ptr(a_planes), T, R, K, N, C.byref(p), _ptr(w_packed), _ptr(spk), R * Nw, _stream()),
               "snn_spike_gemm_lif_mx")
    return spk


def conv3x3_lif_mx(enc: torch.Tensor, shapes, C_in: int, C_out: int, p: snn_params, w_packed: torch.Tensor) -> torch.Tensor:
    """enc int32 [T, Pp, Cw]: planes over levels `shapes` WITH a one-position zero halo around every image
    (Pp = sum n (h+2) (w+2)) -> shared-LIF spike planes int32 [T, P, Nw] (fp4 x fp6 path)"""
    _need_gpu(enc, "enc planes")
    lib = _lib.load()
    T, Pp, Cw = enc.shape
    P = sum(n * h * w for n, h, w in shapes)
    assert Pp == sum(n * (h + 2) * (w + 2) for n, h, w in shapes)
    lv = (snn_rpn_level * len(shapes))(*[snn_rpn_level(None, n, h, w, 0) for n, h, w in shapes])
    Nw = cdiv(C_out, 32)
    spk = torch.empty((T, P, Nw), dtype=torch.int32, device=enc.device)
    _lib.check(lib.snn_conv3x3_lif_mx(_ptr(enc), Pp * Cw, lv, len(shapes), C_in, C_out, T, C.byref(p), _ptr(w_packed), _ptr(spk),
                                      P * Nw, _stream()), "snn_conv3x3_lif_mx")
    return spk


def spike_conv3x3_mx(enc: torch.Tensor, shapes, C_in: int, C_out: int, w_packed: torch.Tensor) -> torch.Tensor:
    """enc int32 [T, Pp, Cw] zero-halo planes over levels `shapes` -> cur fp32 [T, P, Np] (fp4 x fp6 path)"""
    _need_gpu(enc, "enc planes")
    lib = _lib.load()
    T, Pp, Cw = enc.shape
    P = sum(n * h * w for n, h, w in shapes)
    assert Pp == sum(n * (h + 2) * (w + 2) for n, h, w in shapes)
    lv = (snn_rpn_level * len(shapes))(*[snn_rpn_level(None, n, h, w, 0) for n, h, w in shapes])
    Np = cdiv(C_out, 32) * 32
    cur = torch.empty((T, P, Np), dtype=torch.float32, device=enc.device)
    _lib.check(lib.snn_spike_conv3x3_mx(_ptr(enc), Pp * Cw, lv, len(shapes), C_in, C_out, T, _ptr(w_packed), _ptr(cur), Np,
                                        _stream()), "snn_spike_conv3x3_mx")
    return cur


def pad_planes(planes: torch.Tensor, shapes) -> torch.Tensor:
    """[T, P, Cw] spike words over levels `shapes` = [(N,H,W), ...] -> [T, Pp, Cw] with a one-position zero halo around every
    image (Pp = sum N (H+2) (W+2)): the input format of the 3x3 convolution kernels (the head's own encoder writes it directly)"""
    T, _, Cw = planes.shape
    out, pos = [], 0
    for n, h, w in shapes:
        blk = planes[:, pos:pos + n * h * w].reshape(T, n, h, w, Cw)
        out.append(torch.nn.functional.pad(blk, (0, 0, 1, 1, 1, 1)).reshape(T, n * (h + 2) * (w + 2), Cw))
        pos += n * h * w
    return torch.cat(out, dim=1).contiguous()


def _conv_planes(enc: torch.Tensor, shapes):
    """the conv kernels read zero-halo planes; un-padded planes (what encode_nchw returns) are padded here (stage-level calls only)"""
    P = sum(n * h * w for n, h, w in shapes)
    Pp = sum(n * (h + 2) * (w + 2) for n, h, w in shapes)
    if enc.shape[1] == P and P != Pp:
        enc = pad_planes(enc, shapes)
    assert enc.shape[1] == Pp, (enc.shape, P, Pp)
    return enc.contiguous(), P, Pp


def conv3x3_lif_bf16x3(enc: torch.Tensor, shapes, C_in: int, C_out: int, p: snn_params, w_packed: torch.Tensor) -> torch.Tensor:
    """enc int32 [T, P or Pp, Cw] over levels `shapes` = [(N,H,W), ...] -> shared-LIF spike planes int32 [T, P, Nw]"""
    _need_gpu(enc, "enc planes")
    lib = _lib.load()
    enc, P, Pp = _conv_planes(enc, shapes)
    T, _, Cw = enc.shape
    lv = (snn_rpn_level * len(shapes))(*[snn_rpn_level(None, n, h, w, 0) for n, h, w in shapes])
    Nw = cdiv(C_out, 32)
    spk = torch.empty((T, P, Nw), dtype=torch.int32, device=enc.device)
    _lib.check(lib.snn_conv3x3_lif_bf16x3(_ptr(enc), Pp * Cw, lv, len(shapes), C_in, C_out, T, C.byref(p), _ptr(w_packed),
                                          _ptr(spk), P * Nw, _stream()), "snn_conv3x3_lif_bf16x3")
    return spk


def spike_conv3x3_bf16x3(enc: torch.Tensor, shapes, C_in: int, C_out: int, w_packed: torch.Tensor) -> torch.Tensor:
    """enc int32 [T, P or Pp, Cw] over levels `shapes` = [(N,H,W), ...] -> cur fp32 [T, P, Np]"""
    _need_gpu(enc, "enc planes")
    lib = _lib.load()
    enc, P, Pp = _conv_planes(enc, shapes)
    T, _, Cw = enc.shape
    lv = (snn_rpn_level * len(shapes))(*[snn_rpn_level(None, n, h, w, 0) for n, h, w in shapes])
    Np = cdiv(C_out, 32) * 32
    cur = torch.empty((T, P, Np), dtype=torch.float32, device=enc.device)
    _lib.check(lib.snn_spike_conv3x3_bf16x3(_ptr(enc), Pp * Cw, lv, len(shapes), C_in, C_out, T, _ptr(w_packed), _ptr(cur),
                                            Np, _stream()), "snn_spike_conv3x3_bf16x3")
    return cur



def affine_act_nchw(x: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, residual: Optional[torch.Tensor] = None,
                    relu: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """relu?((x * scale[c] + bias[c]) (+ residual)) over NCHW in one pass - FrozenBatchNorm2d / Bottleneck tail of the stock
    backbone, bit-identical to the separate torch launches (snn_hip.h).  ``out`` may be ``x`` (in place)."""
    _need_gpu(x, "affine input")
    lib = _lib.load()
    if x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 4:
        raise _lib.SnnHipError("affine_act_nchw: x must be a contiguous fp32 NCHW tensor (got %s, %s)" % (x.dtype, tuple(x.shape)))
    N, Cc, H, W = x.shape
    scale, bias = _f32c(scale).reshape(-1), _f32c(bias).reshape(-1)
    if scale.numel() != Cc or bias.numel() != Cc:
        raise _lib.SnnHipError("affine_act_nchw: scale / bias need %d channels" % Cc)
    if residual is not None and (residual.shape != x.shape or residual.dtype != torch.float32 or not residual.is_contiguous()):
        raise _lib.SnnHipError("affine_act_nchw: residual must match x (contiguous fp32)")
    y = torch.empty_like(x) if out is None else out
    if x.numel():
        _lib.check(lib.snn_affine_act_nchw(_ptr(x), _ptr(scale), _ptr(bias), _ptr(residual), N, Cc, H * W, int(bool(relu)), _ptr(y),
                                           _stream()), "snn_affine_act_nchw")
    return y


def encode_nchw(feat: torch.Tensor, T: int, p: snn_params) -> torch.Tensor:
    """[N,C,H,W] fp32 -> spike bit-planes uint32 viewed as int32 [T, N*H*W, Cw]"""
    _need_gpu(feat, "feature map")
    lib = _lib.load()
    feat = _f32c(feat)
    N, Cc, H, W = feat.shape
    Cw = cdiv(Cc, 32)
    planes = torch.empty((T, N * H * W, Cw), dtype=torch.int32, device=feat.device)
    _lib.check(lib.snn_encode_nchw(_ptr(feat), N, Cc, H, W, T, C.byref(p), _ptr(planes), N * H * W * Cw, _stream()),
               "snn_encode_nchw")
    return planes


def encode_rows(x: torch.Tensor, T: int, p: snn_params) -> torch.Tensor:
    _need_gpu(x, "x")
    lib = _lib.load()
    x = _f32c(x)
    R, D = x.shape
    Dw = cdiv(D, 32)
    planes = torch.empty((T, R, Dw), dtype=torch.int32, device=x.device)
    _lib.check(lib.snn_encode_rows(_ptr(x), R, D, T, C.byref(p), _ptr(planes), R * Dw, _stream()), "snn_encode_rows")
    return planes


def conv3x3_lif(enc: torch.Tensor, N: int, C_in: int, C_out: int, H: int, W: int, p: snn_params,
                w_packed: torch.Tensor, want_counts: bool = False, want_currents: bool = False):
    """returns spk planes [T, N*H*W, Nw] (+ counts [N]) (+ input currents [T, N*H*W, Nw*32])"""
    _need_gpu(enc, "enc planes")
    lib = _lib.load()
    T = enc.shape[0]
    Nw = cdiv(C_out, 32)
    spk = torch.empty((T, N * H * W, Nw), dtype=torch.int32, device=enc.device)
    counts = torch.zeros((N,), dtype=torch.int64, device=enc.device) if want_counts else None
    cur = torch.zeros((T, N * H * W, Nw * 32), dtype=torch.float32, device=enc.device) if want_currents else None
    _lib.check(lib.snn_conv3x3_lif(_ptr(enc), enc.shape[1] * enc.shape[2], N, C_in, C_out, H, W, T, C.byref(p),
                                   _ptr(w_packed), _ptr(spk), N * H * W * Nw, _ptr(counts), _ptr(cur), _stream()),
               "snn_conv3x3_lif")
    out = (spk,) + ((counts,) if want_counts else ()) + ((cur,) if want_currents else ())
    return out if len(out) > 1 else spk


def spike_gemm(a_rows: torch.Tensor, K: int, N: int, w_packed: torch.Tensor) -> torch.Tensor:
    """a_rows int32 [M, Kw] -> cur fp32 [M, Np]"""
    _need_gpu(a_rows, "spike rows")
    lib = _lib.load()
    M = a_rows.shape[0]
    Np = cdiv(N, 32) * 32
    cur = torch.empty((M, Np), dtype=torch.float32, device=a_rows.device)
    _lib.check(lib.snn_spike_gemm(_ptr(a_rows), M, K, N, _ptr(w_packed), _ptr(cur), Np, _stream()), "snn_spike_gemm")
    return cur


def lif_scan(cur: torch.Tensor, N: int, p: snn_params, want_counts: bool = False):
    """cur fp32 [T, R, ldc] -> planes int32 [T, R, Nw]"""
    _need_gpu(cur, "currents")
    lib = _lib.load()
    cur = _f32c(cur)
    T, R, ldc = cur.shape
    Nw = cdiv(N, 32)
    spk = torch.empty((T, R, Nw), dtype=torch.int32, device=cur.device)
    counts = torch.zeros((R,), dtype=torch.int32, device=cur.device) if want_counts else None
    _lib.check(lib.snn_lif_scan(_ptr(cur), T, R, N, ldc, C.byref(p), _ptr(spk), R * Nw, _ptr(counts), _stream()),
               "snn_lif_scan")
    return (spk, counts) if want_counts else spk


def det_exchange_payload(class_logits: torch.Tensor, box_regression: torch.Tensor, n_images: int, max_det: int = 100):
    """the rows each rank hands to the batch's one all-gather (dp.all_gather_detection_tensors): per image the `max_det`
    RoIs with the highest foreground score as (4 regression values of the best class, score, label), by decreasing score.
    class_logits [N*R, K], box_regression [N*R, 4K] -> payload [N, max_det, 6] fp32, counts [N] int32 (one launch)"""
    _need_gpu(class_logits, "class logits")
    lib = _lib.load()
    RN, K = class_logits.shape
    if RN % n_images or tuple(box_regression.shape) != (RN, 4 * K):
        raise _lib.SnnHipError("det_exchange_payload: %d rows for %d images / box_regression %s" % (RN, n_images, tuple(box_regression.shape)))
    cls = class_logits.contiguous().float()
    reg = box_regression.contiguous().float()
    payload = torch.empty((n_images, max_det, 6), dtype=torch.float32, device=cls.device)
    counts = torch.empty((n_images,), dtype=torch.int32, device=cls.device)
    _lib.check(lib.snn_det_exchange_payload(_ptr(cls), _ptr(reg), n_images, RN // n_images, K, max_det, _ptr(payload),
                                            _ptr(counts), _stream()), "snn_det_exchange_payload")
    return payload, counts


def li_heads(spk: torch.Tensor, K: int, w_heads_packed: torch.Tensor, NA: int, NB: int, p: snn_params,
             want_sums: bool = False):
    _need_gpu(spk, "spike planes")
    lib = _lib.load()
    T, M, Kw = spk.shape
    dev = spk.device
    out_a = torch.empty((M, NA), dtype=torch.float32, device=dev)
    out_b = torch.empty((M, NB), dtype=torch.float32, device=dev)
    sum_a = torch.empty_like(out_a) if want_sums else None
    sum_b = torch.empty_like(out_b) if want_sums else None
    _lib.check(lib.snn_li_heads(_ptr(spk), M * Kw, T, M, K, _ptr(w_heads_packed), NA, NB, C.byref(p), _ptr(out_a),
                                _ptr(out_b), _ptr(sum_a), _ptr(sum_b), _stream()), "snn_li_heads")
    return (out_a, out_b, sum_a, sum_b) if want_sums else (out_a, out_b)


# ---------------------------------------------------------------------------------------------
# whole heads
# ---------------------------------------------------------------------------------------------
class _Workspace:
    """grow-only scratch buffer per (device, stream) (the C ABI never allocates).  Calls on one stream reuse it in
    stream order; calls on different streams get different buffers, so they may overlap."""
    def __init__(self):
        self.buf = {}

    def get(self, device: torch.device, nbytes: int) -> torch.Tensor:
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b


_WS = _Workspace()
_WS_RATES = _Workspace()                     # partial sums of snn_rpn_rates (must not alias the heads' workspace)


def rpn_head_forward(feats: Sequence[torch.Tensor], C_: int, A: int, T: int, p: snn_params,
                     w_shared_packed: torch.Tensor, w_heads_packed: torch.Tensor, spike_rates: bool = False,
                     stage_mask: int = 7):
    """Returns (out_logits [P,A], out_bbox [P,4A], level_rows, extras) — position-major outputs.
    ``stage_mask`` (SNN_STAGE_*: 1 encode, 2 conv+LIF, 4 LI heads) is for profiling only."""
    lib = _lib.load()
    if len(feats) == 0 or len(feats) > _lib.SNN_MAX_LEVELS:
        raise _lib.SnnHipError("RPN head takes 1..%d feature levels, got %d" % (_lib.SNN_MAX_LEVELS, len(feats)))
    feats = [_f32c(f) for f in feats]
    for f in feats:
        _need_gpu(f, "feature map")
        if f.dim() != 4 or f.shape[1] != C_:
            raise _lib.SnnHipError("feature map must be [N,%d,H,W], got %s" % (C_, tuple(f.shape)))
    dev = feats[0].device
    lv = (snn_rpn_level * len(feats))()
    rows = []
    for l, f in enumerate(feats):
        lv[l] = snn_rpn_level(f.data_ptr(), f.shape[0], f.shape[2], f.shape[3], 0)
        rows.append(f.shape[0] * f.shape[2] * f.shape[3])
    P = sum(rows)
    max_n = max(f.shape[0] for f in feats)
    ws_bytes = lib.snn_rpn_head_workspace_bytes(lv, len(feats), C_, A, T, p.precision)
    ws = _WS.get(dev, ws_bytes)
    out_logits = torch.empty((P, A), dtype=torch.float32, device=dev)
    out_bbox = torch.empty((P, 4 * A), dtype=torch.float32, device=dev)
    counts = sum_l = sum_b = rates = None
    if spike_rates:
        counts = torch.empty((len(feats), max_n), dtype=torch.int64, device=dev)
        sum_l = torch.empty_like(out_logits)
        sum_b = torch.empty_like(out_bbox)
    _lib.check(lib.snn_rpn_head_forward_stages(lv, len(feats), C_, A, T, C.byref(p), _ptr(w_shared_packed),
                                               _ptr(w_heads_packed), _ptr(out_logits), _ptr(out_bbox),
                                               _ptr(counts), _ptr(sum_l), _ptr(sum_b), _ptr(ws), ws.numel(),
                                               int(stage_mask), _stream()),
               "snn_rpn_head_forward")
    if spike_rates and stage_mask == 7:
        # the finished [.., 2] = (rate, FLOPs) rows of rpn.py:171-195: two small launches, no torch arithmetic
        rates = torch.empty((len(feats), 3, max_n, 2), dtype=torch.float32, device=dev)
        rws = _WS_RATES.get(dev, lib.snn_rpn_rates_workspace_bytes(len(feats), max_n))
        _lib.check(lib.snn_rpn_rates(lv, len(feats), C_, A, T, _ptr(counts), _ptr(sum_l), _ptr(sum_b), _ptr(rates), _ptr(rws),
                                     rws.numel(), _stream()), "snn_rpn_rates")
    return out_logits, out_bbox, rows, (counts, sum_l, sum_b, rates)


def det_head_forward(x: torch.Tensor, Hd: int, K: int, K4: int, T: int, p: snn_params, w6_packed: torch.Tensor,
                     w7_packed: torch.Tensor, w_heads_packed: torch.Tensor, spike_rates: bool = False, w6_inner: int = 0):
    lib = _lib.load()
    _need_gpu(x, "box features")
    x = _f32c(x).flatten(1)
    if x.data_ptr() % 16:                                        # (a view at an odd offset: the word-major encoder loads 16-byte pieces)
        x = x.clone()
    R, D = x.shape
    dev = x.device
    out_cls = torch.empty((R, K), dtype=torch.float32, device=dev)
    out_bbox = torch.empty((R, K4), dtype=torch.float32, device=dev)
    c6 = c7 = s_c = s_b = None
    if spike_rates:
        c6 = torch.empty((R,), dtype=torch.int32, device=dev)
        c7 = torch.empty((R,), dtype=torch.int32, device=dev)
        s_c = torch.empty_like(out_cls)
        s_b = torch.empty_like(out_bbox)
    if R == 0:
        return out_cls, out_bbox, (c6, c7, s_c, s_b)
    ws_bytes = lib.snn_det_head_workspace_bytes(R, D, Hd, K, K4, T, p.precision)
    ws = _WS.get(dev, ws_bytes)
    _lib.check(lib.snn_det_head_forward_k(_ptr(x), R, D, Hd, K, K4, T, C.byref(p), _ptr(w6_packed), int(w6_inner), _ptr(w7_packed),
                                          _ptr(w_heads_packed), _ptr(out_cls), _ptr(out_bbox), _ptr(c6), _ptr(c7),
                                          _ptr(s_c), _ptr(s_b), _ptr(ws), ws.numel(), _stream()),
               "snn_det_head_forward")
    return out_cls, out_bbox, (c6, c7, s_c, s_b)


def det_rates(extras, D: int, Hd: int, K: int, K4: int, T: int, only_one_bbox: bool) -> torch.Tensor:
    """(c6, c7, sum_cls, sum_bbox) of a spike-rate head call -> rates [4, R, 2] = (rate, FLOPs) rows of faster_rcnn.py:568-618"""
    lib = _lib.load()
    c6, c7, s_c, s_b = extras
    R = c6.shape[0]
    rates = torch.empty((4, R, 2), dtype=torch.float32, device=c6.device)
    if R:
        _lib.check(lib.snn_det_rates(R, D, Hd, K, K4, T, int(bool(only_one_bbox)), _ptr(c6), _ptr(c7), _ptr(s_c), _ptr(s_b),
                                     _ptr(rates), _stream()), "snn_det_rates")
    return rates


# ---------------------------------------------------------------------------------------------
# RoIAlign fused with the detector encoder
# ---------------------------------------------------------------------------------------------
def _roi_levels(feats, scales):
    lv = (snn_roi_level * len(feats))()
    keep = []
    for i, (f, sc) in enumerate(zip(feats, scales)):
        _need_gpu(f, "feature map")
        f = _f32c(f)
        keep.append(f)
        lv[i] = snn_roi_level(f.data_ptr(), f.shape[2], f.shape[3], float(sc), 0)
    return lv, keep


def roi_align_encode(feats, scales, rois: torch.Tensor, roi_batch: torch.Tensor, roi_level: torch.Tensor, T: int,
                     p: snn_params, want_pooled: bool = False):
    """feats: list of [N,C,H,W]; rois [R,4]; roi_batch/roi_level int32 [R] -> encoder planes int32 [T, R, Dw]
    (+ the pooled [R, C*49] features when asked: parity tests)"""
    lib = _lib.load()
    lv, keep = _roi_levels(feats, scales)
    Cc = keep[0].shape[1]
    rois = _f32c(rois)
    roi_batch = roi_batch.to(torch.int32).contiguous()
    roi_level = roi_level.to(torch.int32).contiguous()
    R = rois.shape[0]
    Dw = cdiv(Cc * 49, 32)
    planes = torch.empty((T, R, Dw), dtype=torch.int32, device=rois.device)
    pooled = torch.empty((R, Cc * 49), dtype=torch.float32, device=rois.device) if want_pooled else None
    _lib.check(lib.snn_roi_align_encode(lv, len(keep), Cc, _ptr(rois), _ptr(roi_batch), _ptr(roi_level), R, T, C.byref(p),
                                        _ptr(planes), R * Dw, _ptr(pooled), _stream()), "snn_roi_align_encode")
    return (planes, pooled) if want_pooled else planes


def det_head_forward_roialign(feats, scales, rois, roi_batch, roi_level, Hd: int, K: int, K4: int, T: int, p: snn_params,
                              w6_packed, w7_packed, w_heads_packed, spike_rates: bool = False, w6_inner: int = 0):
    lib = _lib.load()
    lv, keep = _roi_levels(feats, scales)
    Cc = keep[0].shape[1]
    rois = _f32c(rois)
    roi_batch = roi_batch.to(torch.int32).contiguous()
    roi_level = roi_level.to(torch.int32).contiguous()
    R, dev = rois.shape[0], rois.device
    out_cls = torch.empty((R, K), dtype=torch.float32, device=dev)
    out_bbox = torch.empty((R, K4), dtype=torch.float32, device=dev)
    c6 = c7 = s_c = s_b = None
    if spike_rates:
        c6 = torch.empty((R,), dtype=torch.int32, device=dev)
        c7 = torch.empty((R,), dtype=torch.int32, device=dev)
        s_c = torch.empty_like(out_cls)
        s_b = torch.empty_like(out_bbox)
    if R == 0:
        return out_cls, out_bbox, (c6, c7, s_c, s_b)
    ws_bytes = lib.snn_det_head_workspace_bytes(R, Cc * 49, Hd, K, K4, T, p.precision)
    ws = _WS.get(dev, ws_bytes)
    _lib.check(lib.snn_det_head_forward_roialign_k(lv, len(keep), Cc, _ptr(rois), _ptr(roi_batch), _ptr(roi_level), R, Hd, K,
                                                   K4, T, C.byref(p), _ptr(w6_packed), int(w6_inner), _ptr(w7_packed), _ptr(w_heads_packed),
                                                   _ptr(out_cls), _ptr(out_bbox), _ptr(c6), _ptr(c7), _ptr(s_c), _ptr(s_b),
                                                   _ptr(ws), ws.numel(), _stream()), "snn_det_head_forward_roialign")
    return out_cls, out_bbox, (c6, c7, s_c, s_b)


# ---------------------------------------------------------------------------------------------
# greedy / batched NMS (glue on either side of the heads)
# ---------------------------------------------------------------------------------------------
def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: Optional[torch.Tensor], iou_threshold: float,
                max_keep: Optional[int] = None) -> torch.Tensor:
    """indices of the kept boxes, by decreasing score; suppression only between boxes of equal idxs"""
    _need_gpu(boxes, "boxes")
    lib = _lib.load()
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    order = scores.argsort(descending=True, stable=True)
    b = _f32c(boxes[order])
    cat = idxs[order].to(torch.int32).contiguous() if idxs is not None else None
    max_keep = n if max_keep is None else min(int(max_keep), n)
    keep = torch.empty((max_keep,), dtype=torch.int32, device=boxes.device)
    n_keep = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    ws = _WS.get(boxes.device, lib.snn_nms_workspace_bytes(n))
    _lib.check(lib.snn_nms_sorted(_ptr(b), _ptr(cat), n, float(iou_threshold), max_keep, _ptr(keep), _ptr(n_keep), _ptr(ws),
                                  ws.numel(), _stream()), "snn_nms_sorted")
    k = int(n_keep.item())
    return order[keep[:k].to(torch.int64)]


def nms_keep_mask(boxes: torch.Tensor, scores: torch.Tensor, idxs: Optional[torch.Tensor], iou_threshold: float):
    """Greedy NMS without a host synchronisation: returns (order, kept) where `order` sorts the boxes by decreasing
    score (stable) and kept[i] says whether box order[i] survives; suppression only between boxes of equal idxs."""
    _need_gpu(boxes, "boxes")
    lib = _lib.load()
    n = boxes.shape[0]
    order = scores.argsort(descending=True, stable=True)
    if n == 0:
        return order, torch.zeros((0,), dtype=torch.bool, device=boxes.device)
    b = _f32c(boxes[order])
    cat = idxs[order].to(torch.int32).contiguous() if idxs is not None else None
    keep = torch.full((n,), n, dtype=torch.int32, device=boxes.device)      # unused slots point at a dummy element
    n_keep = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    ws = _WS.get(boxes.device, lib.snn_nms_workspace_bytes(n))
    _lib.check(lib.snn_nms_sorted(_ptr(b), _ptr(cat), n, float(iou_threshold), n, _ptr(keep), _ptr(n_keep), _ptr(ws),
                                  ws.numel(), _stream()), "snn_nms_sorted")
    kept = torch.zeros((n + 1,), dtype=torch.bool, device=boxes.device)
    kept[keep.to(torch.int64)] = True
    return order, kept[:n]


def rpn_proposals(logits: Sequence[torch.Tensor], deltas: Sequence[torch.Tensor], level_hw, strides, base_anchors,
                  image_sizes, pre_nms_top_n: int, post_nms_top_n: int, nms_thresh: float, score_thresh: float,
                  min_size: float, want_pre: bool = True):
    """RPN proposal selection for a batch (rpn.py:420-499) in six launches.
    logits[l] fp32 [N*H*W, A], deltas[l] fp32 [N*H*W, 4A] (position-major), level_hw[l] = (H, W), strides[l] = (sh, sw),
    base_anchors[l] = [A, 4] cell anchors, image_sizes = [(h, w)] * N.
    Returns (boxes [N, post, 4], scores [N, post], counts [N] int32, pre_boxes [N, K, 4] | None, pre_prob [N, K] | None)
    with rows beyond counts[i] zero; candidates ordered by decreasing score."""
    lib = _lib.load()
    L, N = len(logits), len(image_sizes)
    A = logits[0].shape[1]
    dev = logits[0].device
    _need_gpu(logits[0], "RPN logits")
    keep_alive = []
    lv = (_lib.snn_rpn_post_level * L)()
    for l in range(L):
        lo, de = _f32c(logits[l]), _f32c(deltas[l])
        keep_alive += [lo, de]
        H, W = level_hw[l]
        if lo.shape != (N * H * W, A) or de.shape != (N * H * W, 4 * A):
            raise _lib.SnnHipError("rpn_proposals: level %d has shapes %s / %s" % (l, tuple(lo.shape), tuple(de.shape)))
        lv[l].logits, lv[l].deltas, lv[l].H, lv[l].W = lo.data_ptr(), de.data_ptr(), H, W
        lv[l].stride_h, lv[l].stride_w = float(strides[l][0]), float(strides[l][1])
        ba = base_anchors[l].detach().cpu().to(torch.float32)
        for a in range(A):
            for q in range(4):
                lv[l].base_anchors[a][q] = float(ba[a, q])
    K = lib.snn_rpn_proposals_candidates(lv, L, A, int(pre_nms_top_n))
    if K <= 0:
        raise _lib.SnnHipError("rpn_proposals: bad level description")
    hw = (C.c_float * (2 * N))(*[float(v) for s in image_sizes for v in (s[0], s[1])])
    boxes = torch.empty((N, post_nms_top_n, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((N, post_nms_top_n), dtype=torch.float32, device=dev)
    counts = torch.empty((N,), dtype=torch.int32, device=dev)
    pre_b = torch.empty((N, K, 4), dtype=torch.float32, device=dev) if want_pre else None
    pre_p = torch.empty((N, K), dtype=torch.float32, device=dev) if want_pre else None
    ws = _WS.get(dev, lib.snn_rpn_proposals_workspace_bytes(N, K))
    _lib.check(lib.snn_rpn_proposals(lv, L, N, A, hw, int(pre_nms_top_n), int(post_nms_top_n), float(nms_thresh),
                                     float(score_thresh), float(min_size), _ptr(boxes), _ptr(scores), _ptr(counts),
                                     _ptr(pre_b), _ptr(pre_p), _ptr(ws), ws.numel(), _stream()), "snn_rpn_proposals")
    return boxes, scores, counts, pre_b, pre_p


def det_postprocess(class_logits: torch.Tensor, box_regression: torch.Tensor, proposals: torch.Tensor, rois_per_image,
                    image_sizes, box_weights, score_thresh: float, nms_thresh: float, detections_per_img: int,
                    min_size: float = 1e-2):
    """Detection post-processing for a batch (roi_heads.py:1075-1176) in five launches.
    class_logits [R, K], box_regression [R, 4K], proposals [R, 4] (all images concatenated), rois_per_image = [R_i].
    Returns (boxes [N, cap, 4], scores [N, cap], labels [N, cap] int32, counts [N, 2] int32 (fg, bg), all_scores [R, K],
    all_boxes [R, K, 4]); rows of image i: counts[i,0] foreground detections, then counts[i,1] background boxes."""
    lib = _lib.load()
    dev = class_logits.device
    N, K = len(rois_per_image), class_logits.shape[1]
    R = class_logits.shape[0]
    # the kernels index box_regression as [R, 4K] and proposals as [R, 4]: anything else (e.g. the 4-wide output of an
    # only_one_bbox head) would be an out-of-bounds device read
    if tuple(box_regression.shape) != (R, 4 * K) or tuple(proposals.shape) != (R, 4) or sum(int(r) for r in rois_per_image) != R:
        raise _lib.SnnHipError("det_postprocess: class_logits %s needs box_regression [%d, %d] and proposals [%d, 4] "
                               "(got %s, %s; %d RoIs listed per image)" % (tuple(class_logits.shape), R, 4 * K, R, tuple(box_regression.shape),
                                                                          tuple(proposals.shape), sum(int(r) for r in rois_per_image)))
    _need_gpu(class_logits, "class logits")
    lg, dl, pr = _f32c(class_logits), _f32c(box_regression), _f32c(proposals)
    rmax = max(rois_per_image) if N else 0
    cap = int(detections_per_img) + rmax
    boxes = torch.empty((N, cap, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((N, cap), dtype=torch.float32, device=dev)
    labels = torch.empty((N, cap), dtype=torch.int32, device=dev)
    counts = torch.empty((N, 2), dtype=torch.int32, device=dev)
    all_scores = torch.empty((R, K), dtype=torch.float32, device=dev)
    all_boxes = torch.empty((R, K, 4), dtype=torch.float32, device=dev)
    rpi = (C.c_int * N)(*[int(r) for r in rois_per_image])
    hw = (C.c_float * (2 * N))(*[float(v) for s in image_sizes for v in (s[0], s[1])])
    ws = _WS.get(dev, max(1, lib.snn_det_postprocess_workspace_bytes(N, rmax, K)))
    bw = (C.c_float * 4)(*[float(v) for v in box_weights])
    _lib.check(lib.snn_det_postprocess(_ptr(lg), _ptr(dl), _ptr(pr), rpi, N, K, hw, bw, float(score_thresh), float(nms_thresh),
                                       int(detections_per_img), float(min_size), _ptr(all_scores), _ptr(all_boxes),
                                       _ptr(boxes), _ptr(scores), _ptr(labels), _ptr(counts), cap, _ptr(ws), ws.numel(),
                                       _stream()), "snn_det_postprocess")
    return boxes, scores, labels, counts, all_scores, all_boxes


# ---------------------------------------------------------------------------------------------
# device guard: every wrapper above launches on "the current stream of the current device"; a caller holding tensors on
# another GPU of the process (cuda:1 while cuda:0 is current) gets that device made current for the call, as torch's
# own operators do
# ---------------------------------------------------------------------------------------------
def _cuda_device_of(values) -> Optional[torch.device]:
    for a in values:
        if isinstance(a, torch.Tensor):
            if a.is_cuda:
                return a.device
        elif isinstance(a, (list, tuple)):
            d = _cuda_device_of(a)
            if d is not None:
                return d
    return None


def _on_tensor_device(fn):
    import functools

    @functools.wraps(fn)
    def guarded(*args, **kw):
        dev = _cuda_device_of(args) or _cuda_device_of(tuple(kw.values()))
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kw)
        with torch.cuda.device(dev):
            return fn(*args, **kw)
    return guarded


for _name in ("pack_conv3x3", "pack_linear", "pack_heads", "pack_conv3x3_bf16x3", "pack_linear_bf16x3", "pack_linear_mx",
              "pack_conv3x3_mx", "spike_gemm_bf16x3", "spike_gemm_lif_bf16x3", "spike_gemm_mx", "spike_gemm_lif_mx",
              "conv3x3_lif_mx", "spike_conv3x3_mx", "conv3x3_lif_bf16x3", "spike_conv3x3_bf16x3", "affine_act_nchw", "encode_nchw", "encode_rows",
              "conv3x3_lif", "spike_gemm", "lif_scan", "det_exchange_payload", "li_heads", "rpn_head_forward", "det_head_forward",
              "det_rates", "roi_align_encode", "det_head_forward_roialign", "batched_nms", "nms_keep_mask", "rpn_proposals",
              "det_postprocess"):
    globals()[_name] = _on_tensor_device(globals()[_name])
del _name
