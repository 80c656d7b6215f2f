"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads, exports every symbol
include/snn_hip.h declares, the argument validation works without a GPU, and the modules keep the
reference's constructor signatures / state_dict keys."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "snn_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(snn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from snn_automotive_object_detection_amd import _lib
    lib = _lib.load()
    names = _header_symbols()
    assert len(names) >= 25
    assert sorted(_lib.SYMBOLS) == names, (sorted(_lib.SYMBOLS), names)
    for n in names:
        assert hasattr(lib, n)
    assert lib.snn_version() >= 1


def test_size_queries_and_argument_validation_without_gpu():
    from snn_automotive_object_detection_amd import _lib
    lib = _lib.load()
    assert lib.snn_packed_conv3x3_elems(256, 256) == 9 * 8 * 8 * 1024
    assert lib.snn_packed_linear_elems(1024, 12544) == 392 * 32 * 1024
    assert lib.snn_packed_heads_elems(3, 12, 256) == 256 * 16
    lv = (_lib.snn_rpn_level * 1)(_lib.snn_rpn_level(None, 2, 192, 384, 0))
    # 2 plane sets x T x P x Cw x 4 bytes
    assert lib.snn_rpn_head_workspace_bytes(lv, 1, 256, 3, 8, 0) == 2 * 8 * (2 * 192 * 384) * 8 * 4
    assert lib.snn_rpn_head_workspace_bytes(lv, 1, 256, 3, 8, 1) == 2 * 8 * (2 * 192 * 384) * 8 * 4
    assert lib.snn_det_head_workspace_bytes(2000, 12544, 1024, 9, 36, 12, 1) > 0
    # null / bad arguments are rejected before any device work, with a message
    p = _lib.snn_params(0.1, -0.2, 0, 0, 0.25, 0.1, 0, 0)
    rc = lib.snn_encode_rows(None, 4, 4, 8, C.byref(p), None, 0, None)
    assert rc < 0 and b"snn_encode_rows" in lib.snn_last_error()
    rc = lib.snn_det_head_forward(None, 1, 1, 1, 1, 1, 40, C.byref(p), *([None] * 10), 0, None)
    assert rc < 0


def test_module_signatures_and_state_dict_keys():
    import snn_automotive_object_detection_amd as S
    h = S.RPNHeadSNN(256, 3, 8)
    assert sorted(h.state_dict()) == ["conv_bbox.weight", "conv_cls.weight", "shared_conv.weight"]
    assert tuple(h.shared_conv.weight.shape) == (256, 256, 3, 3)
    assert tuple(h.conv_cls.weight.shape) == (3, 256, 1, 1) and tuple(h.conv_bbox.weight.shape) == (12, 256, 1, 1)
    assert h.num_steps == 8 and h.in_channels == 256 and h.num_anchors == 3
    assert float(h.p_enc.v_th) == 0.25                     # custom_utils.py:321-329 reads this
    assert abs(float(h.shared_conv.weight.detach().std()) - 0.01) < 1e-3   # rpn.py:78-82
    d = S.FastRCNNPredictorSNNFull(12544, 1024, 9, 12)
    assert sorted(d.state_dict()) == ["bbox_pred.weight", "cls_score.weight", "fc6.weight", "fc7.weight"]
    assert tuple(d.fc6.weight.shape) == (1024, 12544) and tuple(d.bbox_pred.weight.shape) == (36, 1024)
    d1 = S.FastRCNNPredictorSNNFull(12544, 1024, 9, 12, only_one_bbox=True)
    assert tuple(d1.bbox_pred.weight.shape) == (4, 1024)
    assert float(d.p_enc.v_th) == 0.25 and d.num_steps == 12


def test_product_path_has_no_cpu_fallback():
    import snn_automotive_object_detection_amd as S
    from snn_automotive_object_detection_amd._lib import SnnHipError
    with pytest.raises(SnnHipError):
        S.RPNHeadSNN(32, 3, 4)([torch.randn(1, 32, 4, 4)])
    with pytest.raises(SnnHipError):
        S.FastRCNNPredictorSNNFull(49 * 8, 32, 3, 4)(torch.randn(2, 8, 7, 7))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "snn_automotive_object_detection_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no CPU or eager fallback", ""), fn
