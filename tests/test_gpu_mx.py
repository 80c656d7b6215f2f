"""Block-scaled fp6 digit planes (csrc/snn_mx.h): the packer against its definition, and the spike GEMM on top of it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
P = 6


def _decode(packed: np.ndarray, Kc: int, Np: int):
    """packed uint32 words -> digits [P, Kc, Np, 4, 32] (int), Eb [Kc, Np, 4] (biased exponent)"""
    nx, ny = P * Kc * Np * 16, P * Kc * Np * 8
    X = packed[:nx].reshape(P, Kc, Np, 4, 4)
    Y = packed[nx:nx + ny].reshape(P, Kc, Np, 4, 2)
    S = packed[nx + ny:nx + ny + Kc * Np].view(np.uint8).reshape(Kc, Np, 4)
    frag = np.concatenate([X, Y], axis=-1).astype(np.uint64)              # [..., 6] dwords = 192 bits
    bits = np.zeros(frag.shape[:-1] + (3,), dtype=object)
    digits = np.zeros(frag.shape[:-1] + (32,), dtype=np.int64)
    for j in range(32):
        b = 6 * j
        lo = (frag[..., b >> 5] >> np.uint64(b & 31))
        if (b & 31) > 26:
            lo = lo | (frag[..., (b >> 5) + 1] << np.uint64(32 - (b & 31)))
        code = (lo & np.uint64(63)).astype(np.int64)
        mag = code & 31                                                   # units of 1/8: 0..15 plain, 16 = 2.0
        assert (mag <= 16).all()
        digits[..., j] = np.where(code & 32, -mag, mag)
    return digits, S.astype(np.int64)


def _reconstruct(digits, Eb):
    """w' [Kc, Np, 4, 32] in float64 = sum_p d_p/8 * 2^(Eb - 127 - 5p)"""
    w = np.zeros(digits.shape[1:], dtype=np.float64)
    for p in range(P):
        w += digits[p] / 8.0 * np.exp2((Eb - 127 - 5 * p).astype(np.float64))[..., None]
    return w


def _check(w_ref_blocks, digits, Eb):
    rec = _reconstruct(digits, Eb)
    err = np.abs(rec - w_ref_blocks)
    bound = np.exp2((Eb - 156).astype(np.float64))[..., None]             # half a unit of the last plane
    assert (err <= bound).all()
    mant, ex = np.frexp(w_ref_blocks)                                      # |w| = mant * 2^ex, mant in [0.5, 1): biased exponent = ex + 126
    be = np.where(w_ref_blocks != 0, ex + 126, 0)
    exact_expected = (w_ref_blocks == 0) | (be >= Eb[..., None] - 5)
    assert (err[exact_expected] == 0).all()
    assert (np.abs(digits) <= 16).all()
    return float((err / np.maximum(bound, 1e-300)).max())


def test_pack_linear_mx_definition(gpu_device):
    from snn_automotive_object_detection_amd import ops
    g = torch.Generator().manual_seed(0)
    N, K = 70, 300                                                          # padded: Np = 96, Kc = 3 (K -> 384)
    w = torch.randn(N, K, generator=g) * torch.exp2(torch.randint(-12, 3, (N, K), generator=g).float())
    w[3, :40] = 0.0
    w[5, 64:96] = 0.0                                                       # an all-zero block
    w[7, 0] = 1.9999999
    w[7, 1] = -1.9999999
    w[9, 5] = 1e-41                                                         # denormal
    w[11, :32] = torch.tensor([2.0 ** (-i) for i in range(32)])            # 31 binades in one block
    packed = ops.pack_linear_mx(w.to(gpu_device)).cpu().numpy().view(np.uint32)
    Kc, Np = 3, 96
    assert packed.size == P * Kc * Np * 24 + Kc * Np + 16 and (packed[-16:] == 0).all()
    digits, Eb = _decode(packed, Kc, Np)
    ref = np.zeros((Np, Kc * 128), dtype=np.float64)
    ref[:N, :K] = w.numpy().astype(np.float64)
    blocks = ref.reshape(Np, Kc, 4, 32).transpose(1, 0, 2, 3)
    worst = _check(blocks, digits, Eb)
    assert worst <= 1.0
    assert (digits[:, :, N:] == 0).all()                                    # padding columns are silent


def test_pack_conv3x3_mx_definition(gpu_device):
    from snn_automotive_object_detection_amd import ops
    g = torch.Generator().manual_seed(1)
    Co, Ci = 40, 136                                                        # Cp = 256 -> Kc = 18
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.01
    packed = ops.pack_conv3x3_mx(w.to(gpu_device)).cpu().numpy().view(np.uint32)
    Kc, Np, Cp = 18, 64, 256
    digits, Eb = _decode(packed, Kc, Np)
    ref = np.zeros((Np, 9, Cp), dtype=np.float64)
    ref[:Co, :, :Ci] = w.numpy().astype(np.float64).reshape(Co, Ci, 9).transpose(0, 2, 1)
    blocks = ref.reshape(Np, Kc, 4, 32).transpose(1, 0, 2, 3)
    _check(blocks, digits, Eb)


def _planes(bits: torch.Tensor) -> torch.Tensor:
    """bool [..., K] (K multiple of 32) -> int32 words [..., K/32]"""
    w = (1 << torch.arange(32, dtype=torch.int64))
    v = (bits.reshape(bits.shape[:-1] + (-1, 32)).to(torch.int64) * w).sum(-1)
    return torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32)


@pytest.fixture(params=["4", "8"], ids=["mw4", "mw8"])
def mx_mw(request, monkeypatch):
    """both work-group shapes of k_gemm_mx (8 waves x 64 rows / 4 waves x 128 rows)"""
    monkeypatch.setenv("SNN_MX_MW", request.param)
    return request.param


def test_spike_gemm_mx_repeated_calls_are_stable(gpu_device, mx_mw):
    """regression: the chunk scales are shared by all waves of a work-group; their buffer used to be overwritten one
    micro-step too early, which only showed on warm repeats with >= 3 chunks (lagging waves read the next chunk's scales)"""
    from snn_automotive_object_detection_amd import ops
    g = torch.Generator().manual_seed(7)
    for (M, K, N) in [(512, 384, 64), (1024, 640, 128)]:
        bits = torch.rand(M, K, generator=g) < 0.25
        w = torch.randn(N, K, generator=g) * 0.05
        a = _planes(bits).to(gpu_device)
        wp = ops.pack_linear_mx(w.to(gpu_device))
        ref = bits.double() @ w.double().t()
        for _ in range(6):
            cur = ops.spike_gemm_mx(a, K, N, wp)[:, :N].double().cpu()
            assert float((cur - ref).abs().max()) <= 2e-6


@pytest.mark.parametrize("M,K,N", [(64, 128, 32), (700, 384, 96), (513, 1280, 200)])
def test_spike_gemm_mx_vs_fp64(gpu_device, mx_mw, M, K, N):
    from snn_automotive_object_detection_amd import ops
    g = torch.Generator().manual_seed(M + K)
    bits = torch.rand(M, K, generator=g) < 0.25
    w = torch.randn(N, K, generator=g) * 0.05
    a = _planes(bits).to(gpu_device)
    cur = ops.spike_gemm_mx(a, K, N, ops.pack_linear_mx(w.to(gpu_device)))[:, :N].double().cpu()
    ref = bits.double() @ w.double().t()
    cur3 = ops.spike_gemm_bf16x3(a, K, N, ops.pack_linear_bf16x3(w.to(gpu_device)))[:, :N].double().cpu()
    e_mx, e_b3 = (cur - ref).abs(), (cur3 - ref).abs()
    # as accurate as the exact-product bf16x3 kernel: both only carry the fp32 accumulation error
    assert e_mx.max() <= 2e-6 and e_mx.max() <= 2 * e_b3.max() + 1e-7
    assert float((e_mx ** 2).mean().sqrt()) <= 1.5 * float((e_b3 ** 2).mean().sqrt()) + 1e-9


@pytest.mark.parametrize("T,R,K,N", [(8, 100, 256, 64), (12, 45, 384, 70), (5, 130, 128, 200)])
def test_spike_gemm_lif_mx_vs_unfused(gpu_device, mx_mw, T, R, K, N):
    """linear layer + LIF fused in the row tile == GEMM on the same path followed by the LIF scan (bit for bit)"""
    from snn_automotive_object_detection_amd import ops
    from snn_automotive_object_detection_amd.ops import LIFParameters
    g = torch.Generator().manual_seed(T * R)
    bits = torch.rand(T, R, K, generator=g) < 0.3
    w = torch.randn(N, K, generator=g) * 0.08
    p = ops.make_params(LIFParameters(v_th=torch.tensor(0.25)), LIFParameters(alpha=100, v_th=torch.tensor(0.1)))
    a = _planes(bits).to(gpu_device)
    wp = ops.pack_linear_mx(w.to(gpu_device))
    fused = ops.spike_gemm_lif_mx(a, K, N, p, wp)
    cur = ops.spike_gemm_mx(a.view(T * R, -1), K, N, wp)
    ref = ops.lif_scan(cur.view(T, R, -1), N, p)
    assert torch.equal(fused, ref[..., :fused.shape[-1]])
    assert fused.ne(0).any()


def _pad_planes(planes: torch.Tensor, shapes) -> torch.Tensor:
    """[T, P, Cw] spike words over levels `shapes` -> the same with a one-position zero halo around every image"""
    T, _, Cw = planes.shape
    out, pos = [], 0
    for n, h, w in shapes:
        blk = planes[:, pos:pos + n * h * w].reshape(T, n, h, w, Cw)
        pad = torch.zeros((T, n, h + 2, w + 2, Cw), dtype=planes.dtype)
        pad[:, :, 1:h + 1, 1:w + 1] = blk
        out.append(pad.reshape(T, n * (h + 2) * (w + 2), Cw))
        pos += n * h * w
    return torch.cat(out, dim=1).contiguous()


@pytest.mark.parametrize("C_in,C_out,T,shapes", [
    (128, 64, 8, [(1, 7, 9), (2, 3, 4)]),
    (256, 200, 4, [(2, 16, 12), (1, 1, 1)]),
    (384, 32, 12, [(1, 5, 6)]),
])
def test_conv3x3_mx_vs_fp64_and_fused(gpu_device, mx_mw, C_in, C_out, T, shapes):
    import torch.nn.functional as F
    from snn_automotive_object_detection_amd import ops
    from snn_automotive_object_detection_amd.ops import LIFParameters
    g = torch.Generator().manual_seed(C_in + C_out + T)
    P_ = sum(n * h * w for n, h, w in shapes)
    bits = torch.rand(T, P_, C_in, generator=g) < 0.2
    w = torch.randn(C_out, C_in, 3, 3, generator=g) * 0.05
    p = ops.make_params(LIFParameters(v_th=torch.tensor(0.25)), LIFParameters(alpha=100, v_th=torch.tensor(0.1)))
    enc = _pad_planes(_planes(bits), shapes).to(gpu_device)              # the mx conv reads zero-halo planes
    wp = ops.pack_conv3x3_mx(w.to(gpu_device))
    cur = ops.spike_conv3x3_mx(enc, shapes, C_in, C_out, wp)
    pos = 0
    for n, h, wd in shapes:
        xi = bits[:, pos:pos + n * h * wd].double().reshape(T * n, h, wd, C_in).permute(0, 3, 1, 2)
        exp = F.conv2d(xi, w.double(), padding=1).permute(0, 2, 3, 1).reshape(T, n * h * wd, C_out)
        assert (cur[:, pos:pos + n * h * wd, :C_out].double().cpu() - exp).abs().max() <= 5e-6
        pos += n * h * wd
    fused = ops.conv3x3_lif_mx(enc, shapes, C_in, C_out, p, wp)
    assert torch.equal(fused, ops.lif_scan(cur, C_out, p)[..., :fused.shape[-1]])
    assert fused.ne(0).any()
