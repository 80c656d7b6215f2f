"""ORACLE tooling — fixture specs shared by oracle/make_golden.py (which writes the expected
outputs by running the shimmed reference) and by tests/ (which regenerate the SAME inputs and
weights from oracle/portable_rng.py and compare against the committed outputs).

A fixture = a spec (below; plain data) + tests/golden/<name>.npz (expected outputs, bit-packed
per-step spikes).  Inputs/weights are deterministic functions of the spec.
"""
import os

import numpy as np
import torch

from . import portable_rng as PR

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name -> spec.   RPN: C, A, T, shapes, seed, w_std (reference init is N(0, 0.01²), rpn.py:78-82;
# the fixtures use a bell-shaped stand-in with the same std unless noted), feat_std/feat_mean.
RPN_SPECS = {
    "rpn_c16_T8":      dict(C=16, A=3, T=8, shapes=[(2, 16, 12, 24), (2, 16, 6, 12)], seed=101, w_std=0.05),
    "rpn_c256_T8":     dict(C=256, A=3, T=8, shapes=[(2, 256, 12, 24)], seed=102, w_std=0.01),
    "rpn_c256_T8_odd": dict(C=256, A=3, T=8, shapes=[(1, 256, 13, 22), (2, 256, 5, 7)], seed=103, w_std=0.01),
    "rpn_c256_T4":     dict(C=256, A=3, T=4, shapes=[(1, 256, 6, 10)], seed=104, w_std=0.01, feat_std=2.0),
    "rpn_c256_T12":    dict(C=256, A=3, T=12, shapes=[(1, 256, 6, 10)], seed=112, w_std=0.01),
    "rpn_c256_T16":    dict(C=256, A=3, T=16, shapes=[(1, 256, 6, 10)], seed=116, w_std=0.01),
    "rpn_c256_T24":    dict(C=256, A=3, T=24, shapes=[(1, 256, 6, 10)], seed=124, w_std=0.01),
    "rpn_c64_A5_T8":   dict(C=64, A=5, T=8, shapes=[(3, 64, 9, 17), (3, 64, 1, 1)], seed=130, w_std=0.03),
    # a pyramid with the 5 Cityscapes level aspect ratios at 1/8 scale (24x48 ... 2x3)
    "rpn_c256_T8_pyr": dict(C=256, A=3, T=8,
                            shapes=[(2, 256, 24, 48), (2, 256, 12, 24), (2, 256, 6, 12), (2, 256, 3, 6), (2, 256, 2, 3)],
                            seed=140, w_std=0.01),
}

# DET: C (D = C*49), Hd, K, T, R, seed.  Default nn.Linear init is U(+-1/sqrt(fan_in))
# (faster_rcnn.py:447-467 use it unchanged); the fixtures use exactly that distribution.
DET_SPECS = {
    "det_K9_T12":         dict(C=256, Hd=1024, K=9, T=12, R=16, seed=201),
    "det_K11_T8_R37":     dict(C=256, Hd=1024, K=11, T=8, R=37, seed=202),
    "det_small_T16":      dict(C=8, Hd=64, K=5, T=16, R=9, seed=203),
    "det_K9_T24_onebbox": dict(C=256, Hd=1024, K=9, T=24, R=5, seed=204, only_one_bbox=True),
    "det_K9_T12_R130":    dict(C=256, Hd=1024, K=9, T=12, R=130, seed=205),
}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rpn_inputs(spec):
    C, A, s = spec["C"], spec["A"], spec["seed"]
    w_std = spec.get("w_std", 0.01)
    w_shared = _t(PR.normalish((C, C, 3, 3), s * 10 + 1, std=w_std))
    w_cls = _t(PR.normalish((A, C, 1, 1), s * 10 + 2, std=w_std))
    w_bbox = _t(PR.normalish((4 * A, C, 1, 1), s * 10 + 3, std=w_std))
    feats = [_t(PR.normalish(shape, s * 10 + 4 + l, std=spec.get("feat_std", 1.0),
                             mean=spec.get("feat_mean", 0.0)))
             for l, shape in enumerate(spec["shapes"])]
    return feats, w_shared, w_cls, w_bbox


def det_inputs(spec):
    C, Hd, K, R, s = spec["C"], spec["Hd"], spec["K"], spec["R"], spec["seed"]
    D = C * 49
    K4 = 4 if spec.get("only_one_bbox", False) else 4 * K
    b6, b7 = 1.0 / D ** 0.5, 1.0 / Hd ** 0.5
    w6 = _t(PR.uniform((Hd, D), s * 10 + 1, -b6, b6))
    w7 = _t(PR.uniform((Hd, Hd), s * 10 + 2, -b7, b7))
    w_cls = _t(PR.uniform((K, Hd), s * 10 + 3, -b7, b7))
    w_bbox = _t(PR.uniform((K4, Hd), s * 10 + 4, -b7, b7))
    x = _t(PR.normalish((R, C, 7, 7), s * 10 + 5, std=spec.get("feat_std", 1.0)))
    return x, w6, w7, w_cls, w_bbox


def load_expected(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def unpack_spikes(packed: np.ndarray, shape) -> np.ndarray:
    n = int(np.prod(shape))
    return np.unpackbits(packed)[:n].reshape(tuple(int(s) for s in shape)).astype(np.float32)
