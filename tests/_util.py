"""helpers shared by the parity tests: spike bit-plane <-> dense conversion (little-endian bits:
bit b of word w = channel 32*w + b, see include/snn_hip.h)"""
import numpy as np
import torch


def planes_to_dense(planes: torch.Tensor, K: int) -> np.ndarray:
    """int32 [T, M, Kw] -> float32 [T, M, K]"""
    a = planes.detach().cpu().numpy().view(np.uint32)
    T, M, Kw = a.shape
    bits = np.unpackbits(a.view(np.uint8).reshape(T, M, Kw * 4), axis=2, bitorder="little")
    return bits[:, :, :K].astype(np.float32)


def dense_to_planes(z: np.ndarray) -> torch.Tensor:
    """{0,1} array [T, M, K] -> int32 [T, M, Kw] (CPU tensor)"""
    T, M, K = z.shape
    Kw = (K + 31) // 32
    pad = np.zeros((T, M, Kw * 32), dtype=np.uint8)
    pad[:, :, :K] = z.astype(np.uint8)
    words = np.packbits(pad, axis=2, bitorder="little").view(np.uint32).reshape(T, M, Kw)
    return torch.from_numpy(words.view(np.int32).copy())


def nchw_to_rows(z: torch.Tensor) -> np.ndarray:
    """[T, N, C, H, W] -> [T, N*H*W, C] (row = (n*H + y)*W + x)"""
    T, N, C, H, W = z.shape
    return z.permute(0, 1, 3, 4, 2).reshape(T, N * H * W, C).numpy()


# Spike flips at threshold ties.  Two fp32 summation orders of the same contraction agree to ~1e-7; where the oracle's decayed
# membrane sits that close to the threshold the spike can fall either way and the position / RoI then leaves the 1e-4 output
# tolerance (SURVEY.md §7 risk 1).  The number of such positions is Poisson-like, so the budget carries a variance term (round 5;
# VERDICT r4 P-b: `2 + 3 x rate x N` sat within one sigma of the observation at 14 / 17):
#       budget = lambda + 4 sqrt(lambda) + 2,     lambda = rate x neuron-steps x max(1, T / 8)
# with the rates at what rounds 2-4 OBSERVED (profiles/parity_r2.json .. parity_r4.json; no safety factor) - per neuron-step:
#   RPN on N(0,1) pyramids  : 7 - 11 of 196 416 positions at T = 8 (bf16x3 / f32) = 1.7e-8 .. 2.7e-8, 2 - 5 of 87 984 (bdd), 18 mxfp6 (x2:
#                             its digit planes round weights at 2^-29 of the block maximum); 55 at T = 16 = 6.8e-8: a later step carries
#                             more accumulated rounding, hence the factor max(1, T / 8)                                  -> 3.5e-8
#   RPN behind the backbone : 14 of 49 104 and 42 of 196 416 (firing rates are 3x those of N(0,1) inputs), pooled 1.1e-7 -> 1.1e-7
#   detector                : 1 - 3 of 2000 RoIs on N(0,1) at T = 12, 7 mxfp6, 9 at T = 24 = 2e-8 .. 9e-8 (no growth with T seen)  -> 8e-8
#   detector behind the backbone: 4 - 12 of 2000 RoIs since round 4, whose fc6 runs bin-major (k' = bin * C + c) on the structured-sparse
#                             instruction - an order less like the oracle's k-ascending blocks (same-lease A/B: 7, 7 reference order / 9, 11
#                             bin-major dense / 12, 12 bin-major sparse), 8 on average = 1.6e-7                           -> 2.0e-7
# Against round 4's `2 + 3 x rate x N`: the budgets of the N(0,1) pyramids and of the stand-alone detector SHRANK (RPN full size 31 against 33+,
# detector 14 against 17), the two "behind the backbone" budgets GREW (rpn_in_situ at full size 62 -> 73 positions: the variance term; det_in_situ 17 -> 26:
# the bin-major fc6) - a looser gate there, bought with the attribution below and the 3-sigma warning of record_parity.
# Every full-size test also ATTRIBUTES its flips (first_flip_margins): each first differing spike sits within TIE_MARGIN of the
# threshold in the oracle's trace, so a regression that flips spikes away from ties fails whatever the count.
FLIP_RATE = {"rpn_randn": 3.5e-8, "rpn_in_situ": 1.1e-7, "det": 8e-8, "det_in_situ": 2.0e-7}
FLIP_T_GROWTH = {"rpn_randn": True, "rpn_in_situ": True, "det": False, "det_in_situ": False}
PRECISION_FACTOR = {"bf16x3": 1.0, "f32": 1.0, "f32_strict": 1.0, "mxfp6": 2.0}
TIE_MARGIN = 5e-7            # observed: every first flip sits within 7e-8 of the threshold (profiles/parity_r3.json, parity_r4.json)


def flip_lambda(positions: int, channels: int, steps: int, kind: str = "rpn_in_situ", precision: str = "bf16x3") -> float:
    """expected number of positions / RoIs that hold a hidden spike differing from the oracle's (see the table above)"""
    growth = max(1.0, steps / 8.0) if FLIP_T_GROWTH[kind] else 1.0
    return FLIP_RATE[kind] * PRECISION_FACTOR[precision] * channels * steps * positions * growth


def flip_budget(positions: int, channels: int, steps: int, kind: str = "rpn_in_situ", precision: str = "bf16x3") -> float:
    """how many positions / RoIs may hold a hidden spike that differs from the oracle's: expectation + 4 sigma (Poisson) + 2"""
    lam = flip_lambda(positions, channels, steps, kind, precision)
    return lam + 4.0 * lam ** 0.5 + 2.0


def first_flip_margins(spk_got: np.ndarray, spk_exp: np.ndarray, vdec_exp: np.ndarray, theta: float = 0.1):
    """spike trains [T, ...]: for every neuron whose train differs from the oracle's, the oracle's |v_dec - theta| at the
    FIRST differing step (later differences are consequences).  Returns (neurons that differ, their margins, mask of them)."""
    diff = spk_got != spk_exp
    any_diff = diff.any(axis=0)
    first = diff.argmax(axis=0)
    idx = np.nonzero(any_diff)
    margins = np.abs(vdec_exp[(first[idx],) + idx] - theta)
    return int(any_diff.sum()), margins, any_diff


# ---- post-processing fixtures (tests/golden/post_*.npz; oracle/make_golden.py ran the reference's own bodies) ----
BOX_ATOL = 1e-3        # pixels: box coordinates reach 1536, where one fp32 ulp is 1.2e-4; decode has ~6 roundings
SCORE_ATOL = 2e-6      # sigmoid / softmax of identical fp32 logits: library exp() implementations differ in the last ulp


def split_rows(flat: np.ndarray, counts) -> list:
    out, pos = [], 0
    for c in counts:
        out.append(flat[pos:pos + int(c)])
        pos += int(c)
    return out


def assert_same_detections(got_boxes, got_scores, exp_boxes, exp_scores, got_labels=None, exp_labels=None, what=""):
    """same number of rows, same order (by decreasing score; rows whose scores tie within SCORE_ATOL may be permuted),
    boxes within BOX_ATOL, scores within SCORE_ATOL, labels equal"""
    gb, gs = np.asarray(got_boxes, dtype=np.float64).reshape(-1, 4), np.asarray(got_scores, dtype=np.float64).reshape(-1)
    eb, es = np.asarray(exp_boxes, dtype=np.float64).reshape(-1, 4), np.asarray(exp_scores, dtype=np.float64).reshape(-1)
    assert gb.shape[0] == eb.shape[0], "%s: %d rows, expected %d" % (what, gb.shape[0], eb.shape[0])
    if gb.shape[0] == 0:
        return
    gl = np.zeros(gs.shape) if got_labels is None else np.asarray(got_labels, dtype=np.float64).reshape(-1)
    el = np.zeros(es.shape) if exp_labels is None else np.asarray(exp_labels, dtype=np.float64).reshape(-1)
    assert np.abs(gs - es).max() <= SCORE_ATOL, "%s: scores differ by %g" % (what, np.abs(gs - es).max())

    def canon(b, s, l):                      # order inside groups of (nearly) equal scores by label and coordinates
        grp = np.concatenate([[0], np.cumsum(np.abs(np.diff(s)) > SCORE_ATOL)])
        key = np.lexsort((np.round(b[:, 3], 1), np.round(b[:, 2], 1), np.round(b[:, 1], 1), np.round(b[:, 0], 1), l, grp))
        return b[key], l[key]
    gb2, gl2 = canon(gb, es, gl)             # group by the EXPECTED scores on both sides
    eb2, el2 = canon(eb, es, el)
    assert np.array_equal(gl2, el2), "%s: labels differ" % what
    assert np.abs(gb2 - eb2).max() <= BOX_ATOL, "%s: boxes differ by %g" % (what, np.abs(gb2 - eb2).max())


def record_parity(test: str, **values):
    """append observed off-tolerance counts to gpurun_out/parity_r6.jsonl (copied into profiles/parity_r6.json after a GPU
    run): the flip budgets are set on this evidence.  Where a record carries a budget and an observed count it also gets
    lambda (the budget inverted: budget = lambda + 4 sqrt(lambda) + 2), observed / lambda, and `over_3_sigma`; a run above
    lambda + 3 sqrt(lambda) passes but WARNS - ADVICE r5: the rates are observed means without a safety factor, so a tie rate
    that drifts up by ~1.5 x would otherwise go unseen until it crosses 4 sigma (tools/parity_watch.py lists repeat offenders
    over the kept profiles/parity_r*.json)"""
    import json
    import os
    import warnings
    obs = next((values[k] for k in ("positions_off_tolerance", "rois_off_tolerance") if k in values), None)
    if obs is not None and "budget" in values:
        x = -2.0 + (2.0 + float(values["budget"])) ** 0.5          # sqrt(lambda)
        lam = x * x
        values = dict(values, expected=round(lam, 3), observed_over_expected=round(obs / lam, 3) if lam > 0 else None,
                      over_3_sigma=bool(obs > lam + 3.0 * x))
        if values["over_3_sigma"]:
            warnings.warn("parity: %s observed %d flips against an expectation of %.1f (3 sigma = %.1f, budget %.1f)" % (test, obs, lam, lam + 3.0 * x, values["budget"]))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_r6.jsonl")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps({"test": test, **values}) + "\n")
    except OSError:
        pass
