"""The stock backbone's FrozenBatchNorm2d (+ residual) (+ ReLU) in one pass (csrc/snn_affine.h): bit-identical to the separate
torch launches it replaces (torchvision's FrozenBatchNorm2d.forward / Bottleneck.forward, used at faster_rcnn.py:693-694)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _plain(x, scale, bias, residual, relu):
    out = x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)
    if residual is not None:
        out = out + residual
    return torch.relu(out) if relu else out


@pytest.mark.parametrize("shape", [(2, 64, 96, 192), (1, 3, 7, 9), (3, 17, 5, 5), (2, 256, 1, 1), (1, 8, 1, 4100), (2, 5, 33, 31)])
@pytest.mark.parametrize("residual", [False, True])
@pytest.mark.parametrize("relu", [False, True])
def test_affine_act_bitwise_equals_torch_sequence(gpu_device, shape, residual, relu):
    from snn_automotive_object_detection_amd import ops
    g = torch.Generator(device="cpu").manual_seed(sum(shape) + 2 * residual + relu)
    x = torch.randn(shape, generator=g).to(gpu_device)
    scale = (torch.rand(shape[1], generator=g) * 2 - 0.5).to(gpu_device)
    bias = torch.randn(shape[1], generator=g).to(gpu_device)
    r = torch.randn(shape, generator=g).to(gpu_device) if residual else None
    ref = _plain(x, scale, bias, r, relu)
    out = ops.affine_act_nchw(x, scale, bias, r, relu)
    assert torch.equal(out, ref)
    x2 = x.clone()
    assert ops.affine_act_nchw(x2, scale, bias, r, relu, out=x2) is x2 and torch.equal(x2, ref)      # in place


@pytest.mark.parametrize("residual", [False, True])
def test_affine_act_propagates_nan_inf_and_signed_zero_like_torch(gpu_device, residual):
    """torch's relu is threshold(x, 0, 0) - NaN stays NaN, -0.0 becomes +0.0, +inf stays; the fused kernel must not
    turn a diverged backbone into clean zeros"""
    from snn_automotive_object_detection_amd import ops
    x = torch.tensor([float("nan"), float("inf"), float("-inf"), -0.0, 0.0, -1.0, 1.0, 1e-45], device=gpu_device).reshape(1, 1, 2, 4).repeat(1, 2, 1, 1).contiguous()
    scale = torch.tensor([1.0, -1.0], device=gpu_device)
    bias = torch.tensor([0.0, 0.0], device=gpu_device)
    r = torch.zeros_like(x) if residual else None
    for relu in (False, True):
        ref = _plain(x, scale, bias, r, relu)
        out = ops.affine_act_nchw(x, scale, bias, r, relu)
        assert torch.equal(torch.isnan(out), torch.isnan(ref)) and bool(torch.isnan(out).any())
        keep = ~torch.isnan(ref)
        assert torch.equal(out[keep].view(torch.int32), ref[keep].view(torch.int32))          # bit for bit, signed zeros included


def test_frozen_bn_leaves_its_input_alone_unless_told(gpu_device):
    """the in-place write of the fused path is opt-in (the callers inside stock/backbone.py pass fresh conv outputs)"""
    from snn_automotive_object_detection_amd.stock import backbone as B
    bn = B.FrozenBatchNorm2d(8).to(gpu_device)
    bn.bias.fill_(1.0)
    x = torch.randn(1, 8, 5, 5, device=gpu_device)
    x0 = x.clone()
    with torch.no_grad():
        y = bn(x, relu=True)
        assert torch.equal(x, x0) and y.data_ptr() != x.data_ptr()
        y2 = bn(x, relu=True, inplace=True)
        assert y2.data_ptr() == x.data_ptr() and torch.equal(y2, y)


def test_affine_act_rejects_bad_operands(gpu_device):
    from snn_automotive_object_detection_amd import ops, _lib
    x = torch.randn(1, 4, 3, 3, device=gpu_device)
    with pytest.raises(_lib.SnnHipError):
        ops.affine_act_nchw(x.cpu(), torch.ones(4), torch.zeros(4))
    with pytest.raises(_lib.SnnHipError):
        ops.affine_act_nchw(x, torch.ones(5, device=gpu_device), torch.zeros(5, device=gpu_device))
    with pytest.raises(_lib.SnnHipError):
        ops.affine_act_nchw(x.to(memory_format=torch.channels_last).expand(1, 4, 3, 3)[:, :, ::1, :].permute(0, 1, 3, 2), torch.ones(4, device=gpu_device), torch.zeros(4, device=gpu_device))
    with pytest.raises(_lib.SnnHipError):
        ops.affine_act_nchw(x, torch.ones(4, device=gpu_device), torch.zeros(4, device=gpu_device), residual=x[:, :2])


def test_frozen_bn_and_bottleneck_fused_equal_plain(gpu_device, monkeypatch):
    from snn_automotive_object_detection_amd.stock import backbone as B
    torch.manual_seed(5)
    bn = B.FrozenBatchNorm2d(48).to(gpu_device)
    bn.weight.copy_(torch.rand(48) + 0.5); bn.bias.copy_(torch.randn(48)); bn.running_mean.copy_(torch.randn(48)); bn.running_var.copy_(torch.rand(48) + 0.1)
    x = torch.randn(2, 48, 20, 36, device=gpu_device)
    idt = torch.randn_like(x)
    with torch.no_grad():
        monkeypatch.setattr(B, "FUSED_FROZEN_BN", False)
        ref = [bn(x.clone()), bn(x.clone(), relu=True), bn(x.clone(), residual=idt, relu=True)]
        monkeypatch.setattr(B, "FUSED_FROZEN_BN", True)
        got = [bn(x.clone()), bn(x.clone(), relu=True), bn(x.clone(), residual=idt, relu=True)]
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    # buffers edited in place (load_state_dict does that): the cached per-channel constants follow
    with torch.no_grad():
        bn.load_state_dict({"weight": torch.full((48,), 2.0), "bias": torch.zeros(48), "running_mean": torch.zeros(48), "running_var": torch.ones(48)})
        y = bn(x.clone())
        monkeypatch.setattr(B, "FUSED_FROZEN_BN", False)
        assert torch.equal(y, bn(x.clone()))


def test_backbone_fused_matches_plain_within_conv_noise(gpu_device, monkeypatch):
    """whole ResNet-50-FPN: the element-wise parts are bit-identical, MIOpen's convolutions are not run-to-run
    deterministic (split-K igemm solvers), hence a tolerance on the pyramid"""
    from snn_automotive_object_detection_amd.stock import backbone as B
    torch.manual_seed(6)
    net = B.ResNet50FPN().to(gpu_device).eval()
    x = torch.rand(1, 3, 256, 384, device=gpu_device)
    with torch.no_grad():
        monkeypatch.setattr(B, "FUSED_FROZEN_BN", False)
        ref = net(x)
        monkeypatch.setattr(B, "FUSED_FROZEN_BN", True)
        got = net(x)
    for k in ref:
        scale = float(ref[k].abs().max()) + 1e-6
        assert float((ref[k] - got[k]).abs().max()) <= 1e-4 * scale, k
