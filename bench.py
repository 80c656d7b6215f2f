#!/usr/bin/env python3
"""Benchmark of the spiking-heads hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (defaults: 1, 100, 10)

N > 1 works both ways: under torchrun (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py
--gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment, /root/reference/utils.py:268-312's
contract) and stand-alone (`python bench.py --gpus N`: this process starts the N ranks itself as child processes BEFORE
anything touches the GPU, waits for them and passes rank 0's JSON line through; a rank that dies gives a non-zero exit).

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM:
RPNHeadSNN.forward (T_rpn=8) on the 5-level FPN pyramid of a 1024x2048 Cityscapes batch of 2
(768x1536 after the transform: 192x384 ... 12x24, 256 ch) followed by
FastRCNNPredictorSNNFull.forward (T_det=12) on the 2x1000 RoI features [2000,256,7,7], K=9, fp32,
then the exchange payload (top-100 RoIs per image, snn_det_exchange_payload) and, when N>1, the path's one exchange
step (all-gather of those rows over RCCL, dp.py).  Images shard over ranks (weak scaling: every rank runs its own batch
of 2 - contiguous image blocks per rank like the reference's DistributedSampler(shuffle=False), train.py:598-601).

Inputs (`--inputs backbone`, default): the pyramid is the FPN output of the (random-init) ResNet-50-FPN on seeded
`torch.rand(3,1024,2048)` images through the reference's transform, the RoI features are the
7x7 RoIAlign of 1000 seeded boxes per image on that pyramid.  `--inputs randn`: N(0,1) tensors (round-1 behaviour).

Prints ONE JSON line (rank 0): metric images/s + "roofline" (dominant kernel, timed live with HIP events
on the launch stream, on this workload's own encoder planes) + "cpu_baseline" (the oracle on the host cores, bounded sample,
rank 0 at N=1 only) + "extra" (outside the headline timing, N=1 only): a sustained >= 2 s window of the same step, the
end-to-end model (BASELINE config[2]), the BDD per-rank share (config[3]) and the T=16/24 spike-rate stress workload
(config[4]), each with the launch time and roofline fraction of its own conv+LIF kernel.

N > 1 also runs BASELINE.json's config[3] for real as the `dp_e2e` leg: every rank builds create_model("bdd", 11), takes its
contiguous shard of 4 x N seeded rand(3,720,1280) images (4 per GPU; train.py:598-601), runs the WHOLE model and the ranks
all-gather the decoded detections (dp.all_gather_detections: the RCCL counterpart of coco_eval.py:158-177).

--precision bf16x3 (default): both big contractions run on the bf16 matrix cores with an EXACT 3-way bf16
  split of the fp32 weights (spikes are exactly {0,1}; fp32 accumulation; as accurate as the fp32 MFMA chain,
  tools/bf16x3_numerics.hip) - the results are fp32 results, the executed MFMA work is 3x the algorithmic FLOPs.
--precision f32: fp32 matrix cores, 3x3 conv + LIF fused over T (profiles/r1_c_pmc_final.txt).
"""
import argparse
import json
import os
import platform
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LEVELS_CITY = [(192, 384), (96, 192), (48, 96), (24, 48), (12, 24)]   # 768x1536 / (4,8,16,32,64)
LEVELS_BDD = [(192, 344), (96, 172), (48, 86), (24, 43), (12, 22)]    # 720x1280 -> 768x1365 -> canvas 768x1376
C, A, HD = 256, 3, 1024
ROIS_PER_IMG = 1000
PEAK_F32_MFMA_TFLOPS = 157.3                                     # MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0                                   # dense bf16 (no sparsity)
PEAK_MX_MFMA_TFLOPS = 10000.0                                    # dense fp6 / fp4 block-scaled (spec, no sparsity)
PEAK_BF16_SMFMAC_TFLOPS = 5000.0                                 # structured-sparse bf16 (v_smfmac): 2 x the dense peak, dense-equivalent FLOPs
SPARSE_DENSE_PLANES = 2                                          # period planes e_1, e_2 stay on the dense instruction (csrc/snn_sparse.h)

WORKLOADS = {
    "cityscapes": {"levels": LEVELS_CITY, "image": (1024, 2048), "K": 9, "T_rpn": 8, "T_det": 12, "batch": 2, "spike_rates": False,
                   "dataset": "cityscapes", "name": "cityscapes_1024x2048_b2_heads"},
    # BDD 720x1280 -> 768x1376 canvas (SURVEY.md §8 shape table), 4 images per GPU
    "bdd": {"levels": LEVELS_BDD, "image": (720, 1280), "K": 11, "T_rpn": 8, "T_det": 12, "batch": 4, "spike_rates": False,
            "dataset": "bdd", "name": "bdd_720x1280_b4_heads"},
    "stress": {"levels": LEVELS_CITY, "image": (1024, 2048), "K": 9, "T_rpn": 16, "T_det": 24, "batch": 2, "spike_rates": True,
               "dataset": "cityscapes", "name": "cityscapes_1024x2048_b2_heads_T16_T24_spike_rates"},
}
# reference CLI spelling of the step counts (train.py:60-63 `-t-rpn`, `-t-det`; `--rpn-snn --detector-snn` are implied:
# only the spiking heads exist here)
PRECISIONS = ("bf16x3", "f32", "mxfp6")


def algorithmic_flops(wl):
    pos = wl["batch"] * sum(h * w for h, w in wl["levels"])
    conv = pos * 2 * 9 * C * C * wl["T_rpn"]                      # dominant kernel
    rpn = pos * 2 * (9 * C * C + C * A + C * 4 * A) * wl["T_rpn"]
    det = wl["batch"] * ROIS_PER_IMG * 2 * (C * 49 * HD + HD * HD + HD * 5 * wl["K"]) * wl["T_det"]
    return conv, rpn, det


def dead_steps_kept():
    return os.environ.get("SNN_DEAD_STEPS") == "keep"


def kernel_of(precision):
    """(kernel name, matrix-pipe peak TFLOP/s, MFMA work executed per computed time step / algorithmic FLOPs of a step)"""
    if precision == "f32":
        return "k_conv3x3_lif<false>", PEAK_F32_MFMA_TFLOPS, 1.0
    if precision == "mxfp6":
        # fp4 x fp6 block-scaled MFMA, 6 digit planes per weight at 4x the K per instruction: executed work = 6/4 of the
        # bf16-equivalent; peak = 10 PF dense fp6/fp4 (spec)
        return "k_gemm_mx<3, 4>", PEAK_MX_MFMA_TFLOPS, 6.0
    # MODE = G3_CONV_LIF_TILE: 3x3 conv + LIF over T fused in the tile, on the bf16 matrix cores
    return "k_gemm_bf16x3<G3_CONV_LIF_TILE>", PEAK_BF16_MFMA_TFLOPS, 3.0


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle on the host cores (bounded sample), outside every timed region
# ---------------------------------------------------------------------------------------------
def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(leg, repeats=3, warm=True):
    """the oracle (CPU restatement of the reference loops, un-fused torch ops) on the SAME tensors as the GPU leg - the leg's
    backbone-fed pyramid and RoI features and the modules' own weights, copied to the host once: all images of the batch (pyramid, T_rpn)
    + the detector head on their RoIs (T_det); 1 warm-up (levels 1..4 + 256 RoIs: pages the thread pool and allocator in
    without doubling the cost) + `repeats` timed passes, median reported (side legs: one pass).
    Spike-rate workloads (config[4]) run the oracle's spike-rate variants (rpn.py:126-200, faster_rcnn.py:520-618)."""
    import torch
    from oracle import snn_oracle as OR
    wl = leg.wl
    feats = [f.detach().cpu() for f in leg.feats]
    rois = leg.rois.detach().cpu()
    r, d = leg.rpn_head, leg.det_head
    w_s, w_c, w_b = r.shared_conv.weight.detach().cpu(), r.conv_cls.weight.detach().cpu(), r.conv_bbox.weight.detach().cpu()
    w6, w7 = d.fc6.weight.detach().cpu(), d.fc7.weight.detach().cpu()
    wc, wb = d.cls_score.weight.detach().cpu(), d.bbox_pred.weight.detach().cpu()
    threads = torch.get_num_threads()
    rates = bool(wl["spike_rates"])
    times = []
    with torch.no_grad():
        if warm:
            OR.rpn_head_forward(feats[1:], w_s, w_c, w_b, wl["T_rpn"])          # warm-up
            OR.det_head_forward(rois[:256], w6, w7, wc, wb, wl["T_det"])
        for _ in range(repeats):
            t0 = time.perf_counter()
            OR.rpn_head_forward(feats, w_s, w_c, w_b, wl["T_rpn"], spike_rates=rates)
            OR.det_head_forward(rois, w6, w7, wc, wb, wl["T_det"], spike_rates=rates)
            times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    return {"value": round(wl["batch"] / med, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "cpu": cpu_model_name(), "repeats": repeats, "seconds": [round(t, 2) for t in times],
            "inputs": "identical to the GPU leg (copied to the host)",
            "sample": "b=%d: oracle RPN head (5-level pyramid, T=%d) + detector head (%d RoIs, T=%d)%s on the GPU leg's own inputs and "
                      "weights; %s%d repeat%s, median %.1f s per batch" % (
                          wl["batch"], wl["T_rpn"], rois.shape[0], wl["T_det"], ", spike-rate variants" if rates else "",
                          "1 warm-up + " if warm else "behind the headline's warm-up: ", repeats, "" if repeats == 1 else "s", med)}


# ---------------------------------------------------------------------------------------------
# one workload on one device
# ---------------------------------------------------------------------------------------------
class Leg:
    """modules + HBM-resident inputs of one workload"""

    def __init__(self, wl, precision, dev, seed, inputs, backbone_model=None):
        import torch
        import snn_automotive_object_detection_amd as S
        self.wl, self.dev, self.precision = wl, dev, precision
        torch.manual_seed(1234)                                  # same weights on every rank
        self.rpn_head = S.RPNHeadSNN(C, A, wl["T_rpn"]).to(dev)
        self.det_head = S.FastRCNNPredictorSNNFull(C * 49, HD, wl["K"], wl["T_det"]).to(dev)
        self.set_precision(precision)
        self.rpn_head.spike_rates = self.det_head.spike_rates = wl["spike_rates"]
        self.input_note = None
        if inputs == "backbone":
            self.feats, self.rois, self.input_note = self._backbone_inputs(seed, backbone_model)
        elif inputs == "worst_case":
            g = torch.Generator(device="cpu").manual_seed(seed)
            self.feats = [worst_case_tensor((wl["batch"], C, h, w), g, wl["T_rpn"]).to(dev) for h, w in wl["levels"]]
            self.rois = worst_case_tensor((wl["batch"] * ROIS_PER_IMG, C, 7, 7), g, wl["T_det"] - 1).to(dev)
            self.input_note = "synthetic worst case: every four consecutive channels on one encoder period >= 3 (full nibbles in the structured-sparse planes)"
        else:
            g = torch.Generator(device="cpu").manual_seed(seed)
            self.feats = [torch.randn((wl["batch"], C, h, w), generator=g).to(dev) for h, w in wl["levels"]]
            self.rois = torch.randn((wl["batch"] * ROIS_PER_IMG, C, 7, 7), generator=g).to(dev)
            self.input_note = "N(0,1) tensors"

    def set_precision(self, precision):
        self.precision = precision
        self.rpn_head.precision = self.det_head.precision = precision

    def _backbone_inputs(self, seed, model):
        """FPN pyramid of seeded rand images through transform + ResNet-50-FPN (random init, reference hyper-parameters);
        RoI features = MultiScaleRoIAlign of 1000 seeded boxes per image (sizes log-uniform 16..512 px)"""
        import torch
        from snn_automotive_object_detection_amd.stock.roi_align import MultiScaleRoIAlign
        wl, dev = self.wl, self.dev
        g = torch.Generator(device="cpu").manual_seed(seed)
        H, W = wl["image"]
        images = [torch.rand((3, H, W), generator=g).to(dev) for _ in range(wl["batch"])]
        with torch.no_grad():
            il, _ = model.transform(images)
            fmap = model.backbone(il.tensors)
            feats = [f.contiguous() for f in fmap.values()]
            assert [tuple(f.shape[-2:]) for f in feats] == [tuple(l) for l in wl["levels"]], [f.shape for f in feats]
            props = []
            for (h, w) in il.image_sizes:
                size = torch.exp(torch.rand((ROIS_PER_IMG, 2), generator=g) * (6.238 - 2.773) + 2.773)   # 16..512 px
                ctr = torch.rand((ROIS_PER_IMG, 2), generator=g) * torch.tensor([float(w), float(h)])
                b = torch.cat([ctr - size / 2, ctr + size / 2], 1)
                b[:, 0::2] = b[:, 0::2].clamp(0, float(w))
                b[:, 1::2] = b[:, 1::2].clamp(0, float(h))
                props.append(b.to(dev))
            pool = MultiScaleRoIAlign(featmap_names=["0", "1", "2", "3"], output_size=7, sampling_ratio=2)
            rois = pool(fmap, props, il.image_sizes).contiguous()
        del images, fmap
        note = ("FPN output of the random-init ResNet-50-FPN on seeded rand(3,%d,%d) images (transform 768/1536); RoI features = "
                "7x7 RoIAlign of 1000 seeded boxes per image" % (H, W))
        return feats, rois, note

    def step(self):
        from snn_automotive_object_detection_amd import dp, ops
        rpn_out = self.rpn_head(self.feats)
        det_out = self.det_head(self.rois)
        if self.wl["spike_rates"]:       # the spike-rate variants return rate tensors only (faster_rcnn.py:520-618)
            return rpn_out, det_out
        cls, deltas = det_out
        payload, counts = ops.det_exchange_payload(cls, deltas, self.wl["batch"], 100)   # top-100 RoIs per image, one launch
        return dp.all_gather_detection_tensors(payload, counts)  # no-op at world == 1

    def step_local(self):
        """the same step without the collective (side legs that only rank 0 runs)"""
        from snn_automotive_object_detection_amd import ops
        rpn_out = self.rpn_head(self.feats)
        det_out = self.det_head(self.rois)
        if self.wl["spike_rates"]:
            return rpn_out, det_out
        cls, deltas = det_out
        return ops.det_exchange_payload(cls, deltas, self.wl["batch"], 100)

    def exchange_only(self):
        """(payload, counts) of one step, for timing the exchange by itself"""
        from snn_automotive_object_detection_amd import ops
        cls, deltas = self.det_head(self.rois)
        return ops.det_exchange_payload(cls, deltas, self.wl["batch"], 100)

    # ---- per-kernel timing with HIP events on the launch stream ----
    @staticmethod
    def time_ms(fn, iters):
        import torch
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        fn()                                                 # (untimed: a call's first-use costs - allocator growth behind a long side leg - are not its launch time)
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / iters

    def kernel_breakdown(self, iters):
        """conv+LIF launch timed on THIS pyramid's encoder planes: a full RPN-head call leaves them in the workspace, and
        nothing else runs on that workspace until the stage-2 launches are done"""
        from snn_automotive_object_detection_amd import ops
        wl = self.wl
        p = self.rpn_head._params()
        w_sh = self.rpn_head._packed_shared()
        w_hd = self.rpn_head._cache_heads.val
        T = wl["T_rpn"]
        ops.rpn_head_forward(self.feats, C, A, T, p, w_sh, w_hd, stage_mask=7)
        conv_ms = self.time_ms(lambda: ops.rpn_head_forward(self.feats, C, A, T, p, w_sh, w_hd, stage_mask=2), iters)
        from snn_automotive_object_detection_amd import _lib
        self.conv_sparse = bool(_lib.load().snn_debug_last_conv_path())      # planes e_3 .. on the structured-sparse instruction?
        enc_ms = self.time_ms(lambda: ops.rpn_head_forward(self.feats, C, A, T, p, w_sh, w_hd, stage_mask=1), iters)
        rpn_ms = self.time_ms(lambda: self.rpn_head(self.feats), iters)
        det_ms = self.time_ms(lambda: self.det_head(self.rois), iters)
        return {"rpn_head": rpn_ms, "rpn_encode": enc_ms, "rpn_conv3x3_lif": conv_ms, "det_head": det_ms}

    def conv_only_fn(self):
        """the conv + LIF launch alone, on the planes a full head call left in the workspace (as kernel_breakdown times it)"""
        from snn_automotive_object_detection_amd import ops
        p, w_sh, w_hd, T = self.rpn_head._params(), self.rpn_head._packed_shared(), self.rpn_head._cache_heads.val, self.wl["T_rpn"]
        ops.rpn_head_forward(self.feats, C, A, T, p, w_sh, w_hd, stage_mask=7)
        return lambda: ops.rpn_head_forward(self.feats, C, A, T, p, w_sh, w_hd, stage_mask=2)

    def held_clocks(self, step_ms, conv_ms):
        """{step, conv_lif} clocks in GHz held while the whole step / the conv + LIF launch run back to back (None where the probe is unavailable)"""
        try:
            step = held_clock_ghz(self.step_local, step_ms, self.dev)
            conv = held_clock_ghz(self.conv_only_fn(), conv_ms, self.dev)
        except Exception as e:                                    # (an older libsnnhip.so behind SNN_HIP_LIB)
            return {"step_ghz": None, "conv_lif_ghz": None, "error": repr(e)[:200]}
        return {"step_ghz": round(step, 4) if step else None, "conv_lif_ghz": round(conv, 4) if conv else None, "nominal_ghz": NOMINAL_CLOCK_GHZ,
                "method": "one-wave probe on a side stream beside >= 30 ms of back-to-back launches: 0.1 GHz x d(s_memtime) / d(s_memrealtime)"}

    def roofline(self, conv_ms, traffic=None, traffic_source=None):
        conv_fl, _, _ = algorithmic_flops(self.wl)
        kernel, peak, per_step = kernel_of(self.precision)
        T = self.wl["T_rpn"]
        # dead time steps: the conv of the last step cannot reach an output and is not executed, so the launch
        # does (T-1)/T of the dense work SURVEY 8(d) counts.  `achieved` stays the ALGORITHMIC (dense-equivalent) rate =
        # SURVEY's FLOPs of the launch / its duration; `frac` prices the EXECUTED MFMA work against the matrix-pipe peak.
        exec_factor = per_step * (1.0 if (dead_steps_kept() or T < 2) else (T - 1) / T)
        achieved = conv_fl / (conv_ms * 1e-3) / 1e12
        if self.precision == "bf16x3" and getattr(self, "conv_sparse", False):
            # The stage is k_gemm_lif_sparse<true> (round 4 also timed k_compress_planes here; since round 5 the encoder launch compresses).  Of the T - 1 period planes two run on the dense instruction (peak 2.5 PF)
            # and T - 3 on the structured-sparse one (v_smfmac_f32_16x16x64_bf16: 64 k per instruction, peak 5 PF dense-equivalent);
            # frac = (time the executed work takes at those peaks) / (measured time of the whole stage).
            Tc = T - 1
            per_plane = conv_fl / T * 3.0                                   # three bf16 weight planes per period plane
            ex_dense, ex_sparse = per_plane * SPARSE_DENSE_PLANES, per_plane * (Tc - SPARSE_DENSE_PLANES)
            t_peak = ex_dense / (PEAK_BF16_MFMA_TFLOPS * 1e12) + ex_sparse / (PEAK_BF16_SMFMAC_TFLOPS * 1e12)
            return {"bound": "mfma", "kernel": "k_gemm_lif_sparse<true>", "achieved": round(achieved, 2),
                    "peak": round(conv_fl / t_peak / 1e12, 1), "unit": "TFLOP/s", "frac": round(t_peak / (conv_ms * 1e-3), 4),
                    "traffic": traffic, "traffic_source": traffic_source, "launch_ms": round(conv_ms, 4),
                    # (round 5: the encoder launch writes the planes e_3 .. compressed itself - k_compress_planes runs only under SNN_ENC_FOLD=0)
                    "launches_timed": (["k_compress_planes"] if os.environ.get("SNN_ENC_FOLD") == "0" else []) + ["k_gemm_lif_sparse<true>"],
                    "algorithmic_gflop_per_launch": round(conv_fl / 1e9, 1),
                    "executed_dense_tflops": round(ex_dense / (conv_ms * 1e-3) / 1e12, 2), "executed_sparse_tflops": round(ex_sparse / (conv_ms * 1e-3) / 1e12, 2),
                    "mfma_peak_tflops": PEAK_BF16_MFMA_TFLOPS, "smfmac_peak_tflops": PEAK_BF16_SMFMAC_TFLOPS,
                    "period_planes": {"dense": SPARSE_DENSE_PLANES, "sparse": Tc - SPARSE_DENSE_PLANES}, "time_steps_executed": Tc, "time_steps": T,
                    "algorithmic_frac_of_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    "note": "peak = algorithmic FLOPs / (dense planes' work at 2.5 PF + sparse planes' work at 5 PF); the all-dense launch (SNN_SPARSE=0) is the round-3 kernel"}
        return {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2), "peak": round(peak / exec_factor, 1),
                "unit": "TFLOP/s", "frac": round(achieved * exec_factor / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                "launch_ms": round(conv_ms, 4), "algorithmic_gflop_per_launch": round(conv_fl / 1e9, 1),
                "executed_over_algorithmic": round(exec_factor, 4), "executed_tflops": round(achieved * exec_factor, 2),
                "mfma_peak_tflops": peak, "time_steps_executed": (T if (dead_steps_kept() or T < 2) else T - 1), "time_steps": T,
                "algorithmic_frac_of_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4)}

    def spike_stats(self):
        """measured firing rates of this workload's inputs: encoder planes (popcount) and shared-LIF (the head's counts)"""
        import torch
        from snn_automotive_object_detection_amd import ops
        wl = self.wl
        T = wl["T_rpn"]
        p = self.rpn_head._params()
        lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=self.dev)
        enc_bits = enc_n = 0
        for f in self.feats:
            planes = ops.encode_nchw(f, T, p)
            enc_bits += int(lut[planes.view(torch.uint8).to(torch.int64)].sum())
            enc_n += T * f.numel()
        _, _, rows, (counts, _, _, _) = ops.rpn_head_forward(self.feats, C, A, T, p, self.rpn_head._packed_shared(),
                                                          self.rpn_head._cache_heads.get((self.rpn_head.conv_cls.weight, self.rpn_head.conv_bbox.weight), ops.pack_heads),
                                                          spike_rates=True)
        lif_bits = int(counts[:, :wl["batch"]].sum())
        planes = ops.encode_rows(self.rois.flatten(1), wl["T_det"], self.det_head._params())
        det_bits = int(lut[planes.view(torch.uint8).to(torch.int64)].sum())
        return {"rpn_encoder_rate": round(enc_bits / enc_n, 4), "rpn_shared_lif_rate": round(lif_bits / enc_n, 4),
                "det_encoder_rate": round(det_bits / (wl["T_det"] * self.rois.numel()), 4)}


def timed_steps(leg, steps, warmup, fence):
    for _ in range(warmup):
        leg.step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        leg.step()
    fence()
    return time.perf_counter() - t0


NOMINAL_CLOCK_GHZ = 2.4                                          # MI355X_MICROARCH.md chip table: the clock the 2.5 PF / 5 PF peaks are quoted at


def held_clock_ghz(fn, est_ms, dev, window_ms=30.0):
    """the shader clock the chip HOLDS while `fn` runs back to back: a one-wave probe (snn_debug_clock_probe: d s_memtime / d s_memrealtime x 100 MHz,
    MI355X_MICROARCH.md 'DVFS give-back' item 6) on a side stream, started once the device is busy with `fn` and ended while it still is"""
    import torch
    from snn_automotive_object_detection_amd import _lib
    lib = _lib.load()
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    n = max(6, int(window_ms / max(est_ms, 1e-3)))
    torch.cuda.synchronize()
    for _ in range(n // 2 + 1):
        fn()
    _lib.check(lib.snn_debug_clock_probe(out.data_ptr(), int(0.5 * n * est_ms * 1e-3 * 1e8) + 1, side.cuda_stream), "snn_debug_clock_probe")
    for _ in range(n + n // 2 + 2):
        fn()
    torch.cuda.synchronize()
    cyc, ticks = (int(v) for v in out.tolist())
    return 0.1 * cyc / ticks if ticks > 0 else None


def e2e_leg(model, dev, iters=5):
    """BASELINE config[2]: the whole model through create_model on 2 x rand(3,1024,2048) (generalized_rcnn.py:80-122):
    images/s and the stage split, each stage fenced by a device synchronisation"""
    import torch
    g = torch.Generator(device="cpu").manual_seed(7)
    imgs = [torch.rand((3, 1024, 2048), generator=g).to(dev) for _ in range(2)]
    sync = torch.cuda.synchronize
    for _ in range(2):
        out = model(imgs)
    sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = model(imgs)
    sync()
    ms = (time.perf_counter() - t0) / iters * 1e3
    st = [0.0] * 4
    with torch.no_grad():
        for _ in range(iters):
            sync(); a = time.perf_counter()
            il, _ = model.transform(imgs); sync(); b = time.perf_counter()
            feats = model.backbone(il.tensors); sync(); c = time.perf_counter()
            props, _ = model.rpn(il, feats); sync(); d = time.perf_counter()
            model.roi_heads(feats, props, il.image_sizes); sync(); e = time.perf_counter()
            for i, v in enumerate((b - a, c - b, d - c, e - d)):
                st[i] += v * 1e3 / iters
    # The RPN head IN SITU (HIP events around the head's forward inside model(imgs), recorded by module hooks) next to the same head
    # called back to back on the same features: the matrix-core launches run the same number of cycles either way, but behind the
    # stock backbone's lighter kernels the chip holds a lower clock (profiles/r4_in_situ.txt: 2.04 against 2.28 GHz, identical
    # cycles per work-group and L2 hits / misses) - config[2] is quoted with that clock.
    head = model.rpn.head
    evs = []
    h1 = head.register_forward_pre_hook(lambda m, i: evs.append([torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]) or evs[-1][0].record())
    h2 = head.register_forward_hook(lambda m, i, o: evs[-1][1].record())
    with torch.no_grad():
        for _ in range(iters):
            model(imgs)
        sync()
        in_situ = statistics.median(a.elapsed_time(b) for a, b in evs)
        h1.remove(); h2.remove()
        feats_l = list(model.backbone(model.transform(imgs)[0].tensors).values())
        head(feats_l); sync()
        alone = Leg.time_ms(lambda: head(feats_l), 10)
        from snn_automotive_object_detection_amd import ops as _ops
        p_, w_sh, w_hd = head._params(), head._packed_shared(), head._cache_heads.val
        conv_alone = Leg.time_ms(lambda: _ops.rpn_head_forward(feats_l, C, A, int(head.num_steps), p_, w_sh, w_hd, stage_mask=2), 10)
    # two batches in flight: one host thread + one HIP stream each (the forward has host syncs on the per-image proposal /
    # detection counts, so a single thread cannot keep two streams fed; workspaces are per stream, ops._Workspace).  The small
    # kernels of one batch (top-k, NMS lists, RoIAlign, transform) then run beside the big contractions of the other.
    from snn_automotive_object_detection_amd import StreamPipeline
    n_str, per = 2, 12
    pipe = StreamPipeline(model, slots=n_str, device=dev)
    pipe.map([imgs] * (3 * n_str))                  # each stream grows its own allocator pool and workspaces first
    sync()
    t0 = time.perf_counter()
    pipe.map([imgs] * (n_str * per))
    sync()
    ms2 = (time.perf_counter() - t0) / (n_str * per) * 1e3
    return {"workload": "create_model('cityscapes', 9, T_rpn=8, T_det=12) on 2 x rand(3,1024,2048), random init, fp32 backbone (stock MIOpen)",
            "value": round(2 / (ms * 1e-3), 2), "unit": "images/s", "ms_per_batch": round(ms, 3),
            "two_streams": {"value": round(2 / (ms2 * 1e-3), 2), "unit": "images/s", "ms_per_batch": round(ms2, 3),
                            "note": "2 batches in flight (2 host threads x 1 HIP stream each)"},
            "stage_ms": {"transform": round(st[0], 3), "backbone_fpn": round(st[1], 3),
                         "rpn_head_and_proposals": round(st[2], 3), "roi_heads_roialign_dethead_postprocess": round(st[3], 3)},
            "rpn_head_ms": {"in_situ": round(in_situ, 4), "stand_alone": round(alone, 4), "conv_lif_stand_alone": round(conv_alone, 4),
                            "conv_lif_in_situ_estimate": round(in_situ - (alone - conv_alone), 4),
                            "note": "same cycles, lower clock behind the stock backbone (profiles/r4_in_situ.txt)"},
            "detections": [int(d["boxes"].shape[0]) for d in out], "proposals": [int(p.shape[0]) for p in props]}


DP_IMAGES_PER_RANK = 4                                           # BASELINE.json config[3]: b = 32 on 8 GPUs
DP_NUM_CLASSES = 11                                              # BDD (model.py:36-49)


def dp_images(indices, dev):
    """global image i of the config[3] batch = rand(3,720,1280) from its own seed: the same image whatever the rank count"""
    import torch
    out = []
    for i in indices:
        g = torch.Generator(device="cpu").manual_seed(9000 + int(i))
        out.append(torch.rand((3, 720, 1280), generator=g).to(dev))
    return out


def dp_e2e_leg(model, dev, rank, world, fence, iters=3):
    """BASELINE.json config[3], one rank's share: create_model("bdd", 11) on this rank's contiguous block of the 4 x world
    images (DistributedSampler(shuffle=False), train.py:598-601; the inference loop train.py:285-297) followed by the path's one
    exchange: all-gather of the DECODED per-image detections (dp.all_gather_detections; the reference's counterpart is
    coco_eval.py:158-177).  images/s = all ranks' images / max-over-ranks time, exchange included."""
    import torch
    import torch.distributed as dist
    from snn_automotive_object_detection_amd import dp
    n_global = DP_IMAGES_PER_RANK * world
    mine = dp.shard_range(n_global, rank, world)
    imgs = dp_images(mine, dev)

    def one():
        dets = model(imgs)
        return dets, dp.all_gather_detections(dets, max_det=1100, device=dev, images_per_rank=DP_IMAGES_PER_RANK)   # ONE collective per batch
    for _ in range(2):
        dets, gathered = one()
    fence()
    t0 = time.perf_counter()
    for _ in range(iters):
        dets, gathered = one()
    fence()
    dt = time.perf_counter() - t0
    xs = []
    for _ in range(5):                                           # the exchange by itself (pack + collective + unpack)
        fence()
        t1 = time.perf_counter()
        gathered = dp.all_gather_detections(dets, max_det=1100, device=dev, images_per_rank=DP_IMAGES_PER_RANK)
        torch.cuda.synchronize()
        xs.append((time.perf_counter() - t1) * 1e3)
    # the same exchange carrying the reference's FULL eval dicts (all_scores / all_boxes / proposals / objectness too:
    # roi_heads.py:1247-1255, generalized_rcnn.py:125-132; dp.ExtrasSpec) - still one collective, ~0.35 MB per image
    spec = dp.ExtrasSpec(DP_NUM_CLASSES)
    xe = []
    for _ in range(3):
        fence()
        t1 = time.perf_counter()
        full = dp.all_gather_detections(dets, max_det=1100, device=dev, images_per_rank=DP_IMAGES_PER_RANK, extras=spec)
        torch.cuda.synchronize()
        xe.append((time.perf_counter() - t1) * 1e3)
    full_keys = sorted(full[0].keys())
    assert len(full) == n_global and all(torch.equal(a["boxes"], b["boxes"]) for a, b in zip(full, gathered))
    mine0 = mine[0] if len(mine) else 0
    for j, d in enumerate(dets):                                 # this rank's block of the gathered list is its own result, key by key
        for k in d:
            assert torch.equal(full[mine0 + j][k], d[k][:1100] if k in ("boxes", "scores", "labels") else d[k]), k
    if world > 1:
        t = torch.tensor([dt, statistics.median(xs), statistics.median(xe)], dtype=torch.float64, device=dev if dp.backend_name() != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, x_ms, xe_ms = float(t[0]), float(t[1]), float(t[2])
    else:
        x_ms, xe_ms = statistics.median(xs), statistics.median(xe)
    assert len(gathered) == n_global, (len(gathered), n_global)
    n_det = [int(d["boxes"].shape[0]) for d in gathered]
    return {"workload": "bdd_720x1280 create_model('bdd', 11, T_rpn=8, T_det=12), %d images per rank, detections all-gathered" % DP_IMAGES_PER_RANK,
            "value": round(n_global * iters / dt, 3), "unit": "images/s", "ms_per_batch": round(dt / iters * 1e3, 3),
            "global_batch": n_global, "images_per_rank": len(imgs), "exchange_ms": round(x_ms, 4),
            "backend": dp.backend_name(), "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "rows_gathered": len(gathered), "detections_per_image_min_max": [min(n_det), max(n_det)],
            "payload_bytes_per_image": 1101 * 6 * 4,
            "exchange_full_eval_dicts_ms": round(xe_ms, 4), "full_eval_dict_keys": full_keys,
            "full_payload_bytes_per_image": (1101 * 6 + spec.width) * 4}


def worst_case_tensor(shape, gen, T):
    """feature values that put FOUR consecutive channels (a nibble of a plane word) on the SAME period n >= 3, the period cycling over the nibbles
    3, 4, ..: every nibble of the structured-sparse planes e_3 .. that holds a spike holds four, so every (M-tile, 64-k step) of those planes takes
    the secondary instruction (csrc/snn_sparse.h) - the slowest input the sparse launches can see.  Constant-current encoder (v += 0.1 (x - v), spike
    at v > 0.25): period n <=> 0.25 / (1 - 0.9^n) < x <= 0.25 / (1 - 0.9^(n-1)); the value is the middle of that interval (+- 2 % jitter)."""
    import torch
    n_max = max(3, min(T - 1, 7))
    periods = list(range(3, n_max + 1))
    mid = [0.5 * (0.25 / (1 - 0.9 ** n) + 0.25 / (1 - 0.9 ** (n - 1))) for n in periods]
    Cc = shape[1]
    per_ch = torch.tensor([mid[(c // 4) % len(periods)] for c in range(Cc)], dtype=torch.float32)
    x = per_ch.view(1, Cc, *([1] * (len(shape) - 2))).expand(*shape).clone()
    return x * (1.0 + 0.02 * (torch.rand(shape, generator=gen) - 0.5))


def planes_hit_stats(z):
    """z: int64 [T, P, Cw] time-step spike planes (bit b of word w = channel 32 w + b) -> (blocks with a >= 3-spike nibble, blocks, ones, bits) over
    the period planes e_3 .. e_(T-1); block = 16 consecutive positions x one 64-k step (word pair)"""
    import torch
    M1 = 0x11111111
    T = z.shape[0]
    hit = blocks = ones = bits = 0
    before = torch.zeros_like(z[0])
    for n in range(1, T):                                                        # period plane e_n = first spike at step n - 1
        e = z[n - 1] & ~before
        before = before | z[n - 1]
        if n < 3:
            continue
        c = (e & M1) + ((e >> 1) & M1) + ((e >> 2) & M1) + ((e >> 3) & M1)       # spikes per nibble (0 .. 4), one nibble-sized counter each
        ge3 = ((c & (c >> 1)) | (c >> 2)) & M1
        P, Cw = e.shape
        pair = (ge3 != 0).view(P, Cw // 2, 2).any(dim=2)                         # (position, 64-k step)
        blk = pair[:(P // 16) * 16].view(P // 16, 16, Cw // 2).any(dim=1)
        hit += int(blk.sum()); blocks += blk.numel()
        ones += int(sum(((c >> (4 * i)) & 0xf).sum() for i in range(8)))
        bits += e.numel() * 32
    return hit, blocks, ones, bits


def secondary_hit_rate(feats, T, p):
    """fraction of (16 consecutive positions, 64-k step) blocks of the structured-sparse planes e_3 .. e_(T-1) in which some nibble holds >= 3 spikes -
    the blocks whose M-tile issues the structured-sparse instruction a second time (estimated on the centre tap: the conv's other eight taps see the same
    planes shifted by a position), and the mean density of those planes"""
    import torch
    from snn_automotive_object_detection_amd import ops
    tot = [0, 0, 0, 0]
    for f in feats:
        z = ops.encode_nchw(f, T, p).to(torch.int64) & 0xffffffff              # [T, P, Cw] time-step planes
        tot = [a + b for a, b in zip(tot, planes_hit_stats(z))]
    return {"secondary_hit_rate_centre_tap": round(tot[0] / max(tot[1], 1), 4), "sparse_plane_density": round(tot[2] / max(tot[3], 1), 5)}


def density_sweep_leg(leg, make_leg_with, steps, fence):
    """the throughput is data-dependent (how 2:4-compressible the period planes e_3 .. are), so the same heads-only step on three
    input distributions, outside the headline timing: the headline's backbone-fed tensors, N(0,1) tensors, and the synthetic WORST case (worst_case_tensor)"""
    rows = {}
    p = leg.rpn_head._params()
    for name in ("backbone", "randn", "worst_case"):
        l2 = leg if name == "backbone" else make_leg_with(name)
        l2.step = l2.step_local
        dt = timed_steps(l2, steps, 5, fence)
        bd = l2.kernel_breakdown(5)
        rows[name] = {"value": round(l2.wl["batch"] * steps / dt, 2), "unit": "images/s", "ms_per_step": round(dt / steps * 1e3, 4),
                      "conv_lif_launch_ms": round(bd["rpn_conv3x3_lif"], 4), "rpn_head_ms": round(bd["rpn_head"], 4), "det_head_ms": round(bd["det_head"], 4),
                      "conv_path_sparse": bool(getattr(l2, "conv_sparse", False)), **secondary_hit_rate(l2.feats, l2.wl["T_rpn"], p), "inputs": l2.input_note}
        if l2 is not leg:
            del l2
    rows["note"] = ("same weights, same step, %d-step windows; secondary_hit_rate = share of (16 positions x 64 k) blocks of the planes e_3 .. whose M-tile issues a second "
                    "structured-sparse instruction (centre-tap estimate); worst_case = every occupied nibble of those planes full" % steps)
    return rows


def sweep_t_leg(leg, iters=8):
    """throughput vs T over the paper's grid (metrics_for_different_timesteps.py:30-33,360-361: T_rpn 4..12, T_det 8..16): the two
    heads are independent, so 9 + 9 timings.  `steps` = time steps whose contractions are executed (dead ones removed),
    `ms_per_step` relative to the headline T (8 / 12), tile shape and fill as the launcher picks them (snn_debug_tile_shape)."""
    import ctypes as Ct
    from snn_automotive_object_detection_amd import _lib, ops
    lib = _lib.load()
    out32 = (Ct.c_int32 * 12)()
    pos = sum(int(f.shape[0] * f.shape[2] * f.shape[3]) for f in leg.feats)
    R = int(leg.rois.shape[0])
    t_rpn0, t_det0 = leg.rpn_head.num_steps, leg.det_head.num_steps
    res = {"rpn": {}, "det": {}}

    def plan_of(o):                                   # the plan of the launch that runs (structured-sparse or the dense tile)
        if o[8]:
            waves = 4 if o[1] else 8                                          # (o[1]: the FAT shape - four waves of up to 256 registers)
            return {"kernel": "k_gemm_lif_sparse", "period_planes": {"dense": o[9], "sparse": o[10]}, "wave_grid": "%d x %d" % (waves // o[7], o[7]),
                    "fill": round(o[11] / float((waves // o[7]) * o[0]), 4)}       # M-tile slots in use / slots of the wave grid
        return {"kernel": "k_gemm_bf16x3", "fill": round(o[3] * o[4] / o[2], 4)}
    try:
        for T in range(4, 13):
            leg.rpn_head.num_steps = T
            leg.rpn_head(leg.feats)
            ms = leg.time_ms(lambda: leg.rpn_head(leg.feats), iters)
            p_, w_sh, w_hd = leg.rpn_head._params(), leg.rpn_head._packed_shared(), leg.rpn_head._cache_heads.val
            conv_ms = leg.time_ms(lambda: ops.rpn_head_forward(leg.feats, C, A, T, p_, w_sh, w_hd, stage_mask=2), iters)   # the conv+LIF launch alone
            _lib.check(lib.snn_debug_tile_shape(1, pos, C, C, T, 0, 0, out32), "snn_debug_tile_shape")
            assert bool(out32[8]) == bool(lib.snn_debug_last_conv_path()), "snn_debug_tile_shape disagrees with the launch that ran"
            res["rpn"][T] = {"ms": round(ms, 4), "conv_lif_ms": round(conv_ms, 4), "steps": out32[4], "tile_rows": out32[2], "per_tile": out32[3],
                             "work_groups": out32[5], **plan_of(out32)}
        for T in range(8, 17):
            leg.det_head.num_steps = T
            leg.det_head(leg.rois)
            ms = leg.time_ms(lambda: leg.det_head(leg.rois), iters)
            _lib.check(lib.snn_debug_tile_shape(0, R, C * 49, HD, T, 0, 6, out32), "snn_debug_tile_shape")
            assert bool(out32[8]) == bool(lib.snn_debug_last_fc6_path()), "snn_debug_tile_shape disagrees with the launch that ran"
            res["det"][T] = {"ms": round(ms, 4), "steps": out32[4], "tile_rows": out32[2], "per_tile": out32[3],
                             "work_groups": out32[5], "rounds_per_cu_pair": round(out32[5] / 512.0, 3), **plan_of(out32)}
    finally:
        leg.rpn_head.num_steps, leg.det_head.num_steps = t_rpn0, t_det0
    for head, t_ref in (("rpn", 8), ("det", 12)):
        ref = res[head][t_ref]["ms"] / res[head][t_ref]["steps"]
        for T, r in res[head].items():
            r["ms_per_step_rel"] = round(r["ms"] / r["steps"] / ref, 4)
    ref = res["rpn"][8]["conv_lif_ms"] / res["rpn"][8]["steps"]
    for T, r in res["rpn"].items():
        r["conv_lif_ms_per_step_rel"] = round(r["conv_lif_ms"] / r["steps"] / ref, 4)
    worst = max(r["ms_per_step_rel"] for h in ("rpn", "det") for r in res[h].values())
    res["worst_ms_per_step_rel"] = worst
    res["worst_conv_lif_ms_per_step_rel"] = max(r["conv_lif_ms_per_step_rel"] for r in res["rpn"].values())
    res["note"] = ("ms = whole head (encoder + contraction(s) + LI heads); the encoder and the LI heads do not shrink with the step count "
                   "(0.07 + 0.04 ms of the RPN head), which is what a short T pays per step; conv_lif_ms = the matrix-core launch alone")
    res["images_per_s_grid"] = {"T_rpn x T_det": "b=%d: images/s = b / (rpn ms + det ms)" % leg.wl["batch"],
                                "min": round(leg.wl["batch"] / ((res["rpn"][12]["ms"] + res["det"][16]["ms"]) * 1e-3), 1),
                                "headline_T8_T12": round(leg.wl["batch"] / ((res["rpn"][8]["ms"] + res["det"][12]["ms"]) * 1e-3), 1),
                                "max": round(leg.wl["batch"] / ((res["rpn"][4]["ms"] + res["det"][8]["ms"]) * 1e-3), 1)}
    return res


def load_committed_traffic(path, source_digest, workload, precision):
    """(HBM bytes per launch of the dominant kernel, where the figure comes from, the profiled side figures) from the committed PMC summary `path`
    (tools/make_traffic_json.py) - ONLY if that file was collected on the source tree this build comes from: the file carries the `source_digest`
    of its tree and a mismatch gives (None, the two digests, {}) - PMC passes of another tree say nothing about this build's launches"""
    name = os.path.basename(path)
    try:
        with open(path) as f:
            doc = json.load(f)
        if doc.get("source_digest") != source_digest:
            return None, ("profiles/%s was collected on source digest %s, this build is %s: traffic not quoted" % (
                name, str(doc.get("source_digest"))[:12], str(source_digest)[:12])), {}
        e = doc[workload][precision]
        c = e["conv"]
        src = ("profiles/" + name + " (same source digest as this build) <- %s: 2 x FETCH_SIZE (gfx950 tallies 128-B requests at 64 B; calibrated for this access pattern, tools/fetch_calib.hip) "
               "+ WRITE_SIZE, separate rocprofv3 --pmc passes of this launch on this workload, committed - NOT collected by this run" % e["source"])
        prof = {"profiled_launch_ms": round(c["avg_us"] / 1e3, 4), "profiled_hbm_gb_per_s": c.get("hbm_gb_per_s"), "profiled_hbm_frac_of_8tb_s": c.get("hbm_frac_of_8tb_s"),
                "profiled_mfma_busy": c.get("mfma_busy"), "profiled_clock_ghz": c.get("clock_ghz_profiled"), "traffic_over_operands": c.get("traffic_over_operands")}
        if "fc6" in e:
            prof["fc6"] = {k: e["fc6"].get(k) for k in ("kernel", "avg_us", "hbm_bytes_per_launch", "hbm_gb_per_s", "mfma_busy", "clock_ghz_profiled", "traffic_over_operands")}
        # the two encoders are the path's HBM-bound streaming launches ("report GB/s for them separately"): algorithmic bytes
        # (fp32 features in, period planes out) / the profiled launch; their measured traffic is 1.00-1.06 x those bytes
        enc = {k[4:]: {q: e[k].get(q) for q in ("kernel", "avg_us", "algorithmic_hbm_bytes", "hbm_bytes_per_launch", "hbm_gb_per_s", "hbm_frac_of_8tb_s", "hbm_frac_of_6p3tb_s_copy_rate", "traffic_over_operands")}
               for k in ("enc_rpn", "enc_det") if k in e}
        if enc:
            prof["encoders"] = enc
        return c["hbm_bytes_per_launch"], src, prof
    except Exception:
        return None, None, {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # (0.27 s of timed work: the fence around a 20-step window was 1.5 % of it, profiles/r5_warmup_steps.txt)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the legs outside the headline timing (sustained / e2e / bdd / stress / alt precision)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra timing leg with the other precision")
    ap.add_argument("--precision", choices=PRECISIONS, default="bf16x3")
    ap.add_argument("--inputs", choices=["backbone", "randn", "worst_case"], default="backbone")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cityscapes",
                    help="cityscapes = BASELINE.json's headline configuration (default); bdd = config[3] per-rank share "
                         "(720x1280, 4 images per GPU, K=11); stress = config[4] (T=16/24, spike-rate outputs on)")
    ap.add_argument("-t-rpn", "--t-rpn", dest="t_rpn", type=int, default=None, help="RPN time steps (reference CLI: -t-rpn)")
    ap.add_argument("-t-det", "--t-det", dest="t_det", type=int, default=None, help="detector time steps (reference CLI: -t-det)")
    ap.add_argument("--rpn-snn", action="store_true", help="accepted for CLI compatibility (always on)")
    ap.add_argument("--detector-snn", action="store_true", help="accepted for CLI compatibility (always on)")
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--sustain-s", type=float, default=2.0)
    ap.add_argument("--sweep-t", action="store_true", help="only the throughput-vs-T grid (T_rpn 4..12, T_det 8..16) besides the headline")
    ap.add_argument("--no-dp-e2e", action="store_true", help="skip the whole-model data-parallel leg (config[3])")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the held-clock probe (runs under rocprofv3 --pmc, where kernels are serialised and the probe says nothing)")
    args = ap.parse_args()

    # ---- stand-alone multi-rank launch: start the ranks BEFORE anything initialises the GPU in this process ----
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from snn_automotive_object_detection_amd import dp       # (imports torch, opens neither HIP nor torch.cuda)
        n_dev = dp.count_gpus_without_hip()                      # asked in a short-lived child; None: unknown, trust --gpus
        if n_dev is not None and n_dev < args.gpus and "SNN_DP_DEVICE" not in os.environ:
            raise SystemExit("--gpus %d but only %d device(s) visible (a test run of the N-rank path on fewer devices: "
                             "SNN_DIST_BACKEND=gloo SNN_DP_DEVICE=0)" % (args.gpus, n_dev))
        sys.exit(dp.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import torch
    import snn_automotive_object_detection_amd as S
    from snn_automotive_object_detection_amd import dp
    import torch.distributed as dist

    wl = dict(WORKLOADS[args.workload])
    if args.t_rpn:
        wl["T_rpn"] = args.t_rpn
    if args.t_det:
        wl["T_det"] = args.t_det
    rank, local, world = dp.init_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    models = {}

    def model_for(dataset, K):
        if (dataset, K) not in models:                           # the backbone that feeds the heads (and the e2e leg)
            torch.manual_seed(4321)
            models[(dataset, K)] = S.create_model(dataset, K, True, True, 0, False, False, 8, 12).to(dev).eval()
        return models[(dataset, K)]

    def make_leg(w, precision):
        m = model_for(w["dataset"], w["K"]) if args.inputs == "backbone" else None
        return Leg(w, precision, dev, 1000 + rank, args.inputs, m)

    leg = make_leg(wl, args.precision)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def fence_local():                                           # legs that run on one rank only
        torch.cuda.synchronize()

    dt = timed_steps(leg, args.steps, args.warmup, fence)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dp.backend_name() != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms = dt / args.steps * 1e3
    value = world * wl["batch"] * args.steps / dt

    # ---- the exchange step by itself (outside the headline timing) ----
    exchange = {"backend": dp.backend_name(), "ranks": world, "ms": None}
    if world > 1:
        payload, counts = leg.exchange_only()
        fence()
        xs = []
        for _ in range(max(5, args.warmup)):
            fence()
            t0 = time.perf_counter()
            g_payload, g_counts = dp.all_gather_detection_tensors(payload, counts)
            torch.cuda.synchronize()
            xs.append((time.perf_counter() - t0) * 1e3)
        exchange.update(ms=round(statistics.median(xs), 4), ranks=dist.get_world_size(),
                        rows_gathered=int(g_payload.shape[0]), payload_bytes_per_rank=int(payload.numel() * 4 + counts.numel() * 4))
        assert g_payload.shape[0] == world * wl["batch"] and int(g_counts.min()) >= 0

    iters = max(3, min(args.steps, 10))
    bd = leg.kernel_breakdown(iters)
    conv_fl, rpn_fl, det_fl = algorithmic_flops(wl)
    # HBM bytes per launch of the dominant kernel: NOT measured by this run (PMC counters need rocprofv3 around the process) but
    # read from the committed PMC passes of the same launch (separate --pmc FETCH_SIZE / WRITE_SIZE runs, tools/prof_round.sh ->
    # tools/make_traffic_json.py), per workload
    from snn_automotive_object_detection_amd import build as _build
    source_digest = _build.source_digest()                       # content hash of everything libsnnhip.so is built from
    TRAFFIC_JSON = "r6_traffic.json"

    def committed_traffic(workload):
        if args.t_rpn or dead_steps_kept():
            return None, None, {}
        return load_committed_traffic(os.path.join(ROOT, "profiles", TRAFFIC_JSON), source_digest, workload, args.precision)
    traffic, traffic_source, traffic_prof = committed_traffic(args.workload)

    def roofline_of(l, bd_, step_ms, t_, t_src, t_prof):
        """roofline of a leg + the clock the chip held under it: `frac` prices the launch against the peaks at the NOMINAL 2.4 GHz; boxes of this
        pool hold different clocks under the same matrix-core load (MI355X_MICROARCH.md 'DVFS give-back' item 5), so `frac_at_held_clock` =
        frac x 2.4 / held clock is the part of the fraction that is the kernel's and not the box's"""
        rf = {**l.roofline(bd_["rpn_conv3x3_lif"], t_, t_src), **t_prof}
        clk = l.held_clocks(step_ms, bd_["rpn_conv3x3_lif"]) if not args.no_clock_probe else {"step_ghz": None, "conv_lif_ghz": None, "note": "--no-clock-probe"}
        if clk.get("conv_lif_ghz"):
            rf["held_clock_ghz"] = clk["conv_lif_ghz"]
            rf["frac_at_held_clock"] = round(rf["frac"] * NOMINAL_CLOCK_GHZ / clk["conv_lif_ghz"], 4)
            rf["launch_cycles_at_held_clock"] = round(bd_["rpn_conv3x3_lif"] * 1e-3 * clk["conv_lif_ghz"] * 1e9)
        return rf, clk
    roofline, held = roofline_of(leg, bd, ms, traffic, traffic_source, traffic_prof)

    out = {
        "metric": "images/sec (T_rpn=%d,T_det=%d, %dx%d b=%d) spiking RPN+RoI heads forward" % (
            wl["T_rpn"], wl["T_det"], wl["image"][0], wl["image"][1], wl["batch"]),
        "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"f32": "f32", "bf16x3": "f32 (weights as exact bf16x3 split, fp32 accumulate)",
                                        "mxfp6": "f32 (weights as 6 fp6 digit planes with block scales, fp32 accumulate)"}[args.precision],
        "data": "synthetic",
        "config": {"precision": args.precision,
                   "workload": "%s: RPNHeadSNN(T=%d) on 5-level pyramid %dx256x{%dx%d..%dx%d} + FastRCNNPredictorSNNFull(T=%d) on "
                               "%d RoIs x 12544, K=%d; random-init weights; inputs: %s" % (
                                   wl["name"], wl["T_rpn"], wl["batch"], wl["levels"][0][0], wl["levels"][0][1], wl["levels"][-1][0],
                                   wl["levels"][-1][1], wl["T_det"], wl["batch"] * ROIS_PER_IMG, wl["K"], leg.input_note),
                   "global_batch": wl["batch"] * world, "parallelism": "dp%d" % world,
                   "exchange": ("all-gather of per-image detections [100x6] (%s, %d ranks)" % (
                       "RCCL" if exchange["backend"] == "nccl" else exchange["backend"], exchange["ranks"])) if world > 1 else "none"},
        "roofline": roofline,
        "held_clock": held,
        "build": {"source_digest": source_digest, "lib": os.path.relpath(_build.LIB_PATH, ROOT) if not os.environ.get("SNN_HIP_LIB") else os.environ["SNN_HIP_LIB"]},
        "breakdown_ms": {k: round(v, 3) for k, v in bd.items()},
        "heads_tflops": round((rpn_fl + det_fl) / ((bd["rpn_head"] + bd["det_head"]) * 1e-3) / 1e12, 2),
        "exchange": exchange,
    }
    enc_rpn = out["roofline"].get("encoders", {}).get("rpn")
    if enc_rpn and enc_rpn.get("algorithmic_hbm_bytes") and bd["rpn_encode"] > 0:      # (the RPN encoder is also timed live: stage 1 of the head by itself)
        enc_rpn["live_ms"] = round(bd["rpn_encode"], 4)
        enc_rpn["live_gb_per_s"] = round(enc_rpn["algorithmic_hbm_bytes"] / (bd["rpn_encode"] * 1e-3) / 1e9, 1)
    if "SNN_DP_DEVICE" in os.environ and world > 1:
        out["config"]["oversubscribed"] = "%d ranks on device %s (test of the N-rank path, not a scaling measurement)" % (world, os.environ["SNN_DP_DEVICE"])

    # ---- the CPU baseline of the headline (the oracle on the host cores, same tensors): before the side legs, whose own single passes then find the
    # thread pool and the allocator warm
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(leg, args.cpu_repeats)
    elif rank == 0:
        out["cpu_baseline"] = None

    extra = {}

    def side_leg(name, fn):                           # a failing side leg is reported in place; the headline line still prints
        try:
            extra[name] = fn()
        except Exception as e:
            extra[name] = {"error": repr(e)[:500]}
            print("bench.py: side leg %r failed on rank %d: %r" % (name, rank, e), file=sys.stderr)

    run_extra = not args.no_extra and not args.sweep_t
    # ---- config[3] for real, every rank takes part: whole model on this rank's image shard + all-gather of the detections
    if run_extra and not args.no_dp_e2e and not wl["spike_rates"]:
        side_leg("dp_e2e", lambda: dp_e2e_leg(model_for("bdd", 11), dev, rank, world, fence))   # (same seed: same weights on every rank)
    if args.sweep_t and rank == 0 and not wl["spike_rates"]:
        side_leg("t_sweep", lambda: sweep_t_leg(leg))
    # ---- the other BASELINE.json configurations, outside the headline timing (rank 0; at N > 1 the other ranks wait at the
    # final barrier meanwhile)
    if run_extra and rank == 0:
        if world == 1:
            if not wl["spike_rates"]:
                side_leg("input_spike_rates", leg.spike_stats)
            # the same step held for >= sustain_s seconds: the clock the chip sustains, not a burst
            n, t_acc = 0, 0.0
            chunk = max(25, args.steps)
            while t_acc < args.sustain_s:
                t_acc += timed_steps(leg, chunk, 0, fence_local)
                n += chunk
            extra["sustained"] = {"value": round(wl["batch"] * n / t_acc, 3), "unit": "images/s", "steps": n,
                                  "seconds": round(t_acc, 3), "ms_per_step": round(t_acc / n * 1e3, 4)}
            if not wl["spike_rates"]:
                side_leg("t_sweep", lambda: sweep_t_leg(leg))
            if not wl["spike_rates"] and args.inputs == "backbone" and args.precision == "bf16x3":
                side_leg("density_sweep", lambda: density_sweep_leg(leg, lambda kind: Leg(wl, args.precision, dev, 1000 + rank, kind, None), min(args.steps, 30), fence_local))
        if world == 1 and not args.no_alt:
            # the same K steps with the other matrix path of the two big contractions: "mxfp6" = fp4 x fp6 block-scaled MFMA
            # on 6 digit planes per weight (passes the same parity tests; weights are rounded at 2^-28 of their block maximum,
            # so it is NOT the headline)
            alt_prec = "bf16x3" if args.precision == "mxfp6" else "mxfp6"

            def alt_leg():
                leg.set_precision(alt_prec)
                try:
                    dt_alt = timed_steps(leg, args.steps, max(1, args.warmup), fence_local)
                finally:
                    leg.set_precision(args.precision)
                return {"precision": alt_prec, "value": round(wl["batch"] * args.steps / dt_alt, 3), "unit": "images/s",
                        "ms_per_step": round(dt_alt / args.steps * 1e3, 4)}
            side_leg("alt_precision", alt_leg)

        def workload_leg(name):
            w2 = WORKLOADS[name]
            l2 = make_leg(w2, args.precision)
            l2.step = l2.step_local                              # (no collective: the other ranks are not in this leg)
            dt2 = timed_steps(l2, args.steps, max(1, args.warmup), fence_local)
            bd2 = l2.kernel_breakdown(iters)
            t2, t2_src, t2_prof = committed_traffic(name)
            rf2, clk2 = roofline_of(l2, bd2, dt2 / args.steps * 1e3, t2, t2_src, t2_prof)
            res = {"workload": "%s (T_rpn=%d, T_det=%d, b=%d, K=%d%s)" % (w2["name"], w2["T_rpn"], w2["T_det"], w2["batch"], w2["K"],
                                                                   ", spike-rate outputs on" if w2["spike_rates"] else ""),
                   "value": round(w2["batch"] * args.steps / dt2, 3), "unit": "images/s",
                   "ms_per_step": round(dt2 / args.steps * 1e3, 4), "roofline": rf2, "held_clock": clk2,
                   "breakdown_ms": {k: round(v, 3) for k, v in bd2.items()},
                   "kernels_over_step": round((bd2["rpn_head"] + bd2["det_head"]) / (dt2 / args.steps * 1e3), 4)}
            if not args.no_cpu_baseline and world == 1:
                # "report, per config ...": the oracle on THIS leg's tensors too - ONE timed pass (25-50 s of host time) behind the
                # headline's warm-up and repeats; at N = 1 only, like the headline's (at N > 1 the other ranks wait at the final barrier meanwhile)
                res["cpu_baseline"] = cpu_baseline(l2, repeats=1, warm=False)
            return res
        for name in ("bdd", "stress"):
            if name != args.workload:
                side_leg(name, lambda: workload_leg(name))
        if world == 1 and args.workload == "cityscapes":
            side_leg("e2e", lambda: e2e_leg(model_for("cityscapes", 9), dev))
    if extra:
        out["extra"] = extra
    # ---- the side legs' headline numbers once more at the top level and (compact) inside `config`, which the driver keeps
    other = {}
    for name, key in (("sustained", "sustained_img_s"), ("bdd", "bdd_heads_img_s"), ("stress", "stress_heads_img_s"),
                      ("e2e", "e2e_img_s"), ("dp_e2e", "dp_e2e_bdd_img_s"), ("alt_precision", "mxfp6_img_s")):
        v = extra.get(name, {})
        if isinstance(v, dict) and "value" in v:
            other[key] = v["value"]
            out[key] = v["value"]
    for name in ("bdd", "stress"):
        v = extra.get(name, {})
        if isinstance(v, dict) and "roofline" in v:
            other[name + "_conv_frac"] = v["roofline"]["frac"]
            other[name + "_conv_ms"] = v["roofline"]["launch_ms"]
    if "e2e" in extra and "two_streams" in extra["e2e"]:
        other["e2e_two_streams_img_s"] = out["e2e_two_streams_img_s"] = extra["e2e"]["two_streams"]["value"]
    if "dp_e2e" in extra and "exchange_ms" in extra["dp_e2e"]:
        other["dp_e2e_exchange_ms"] = extra["dp_e2e"]["exchange_ms"]
        other["dp_e2e_ranks"] = extra["dp_e2e"]["rccl_ranks"]
        other["dp_e2e_backend"] = extra["dp_e2e"]["backend"]
    ds = extra.get("density_sweep", {})
    if isinstance(ds, dict) and "worst_case" in ds:
        other["worst_case_inputs_img_s"] = out["worst_case_inputs_img_s"] = ds["worst_case"]["value"]
        other["randn_inputs_img_s"] = out["randn_inputs_img_s"] = ds["randn"]["value"]
    if "t_sweep" in extra and "worst_ms_per_step_rel" in extra["t_sweep"]:
        other["t_sweep_worst_ms_per_step_rel"] = extra["t_sweep"]["worst_ms_per_step_rel"]
    if other:
        out["config"]["other_configs"] = other
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
