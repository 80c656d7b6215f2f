#!/usr/bin/env python3
"""Benchmark of the spiking-heads hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM:
RPNHeadSNN.forward (T_rpn=8) on the 5-level FPN pyramid of a 1024x2048 Cityscapes batch of 2
(768x1536 after the transform: 192x384 ... 12x24, 256 ch) followed by
FastRCNNPredictorSNNFull.forward (T_det=12) on the 2x1000 RoI features [2000,256,7,7], K=9, fp32,
then the exchange payload (top-100 RoIs per image, snn_det_exchange_payload) and, when N>1, the path's one exchange
step (all-gather of those rows, dp.py).
Images shard over ranks (weak scaling: every rank runs its own batch of 2).

Prints ONE JSON line (rank 0): metric images/s + "roofline" (dominant kernel, timed live with HIP events
on the launch stream) + "cpu_baseline" (the oracle on the host cores, bounded sample, rank 0 at N=1 only).

--precision bf16x3 (default): both big contractions run on the bf16 matrix cores with an EXACT 3-way bf16
  split of the fp32 weights (spikes are exactly {0,1}; fp32 accumulation; as accurate as the fp32 MFMA chain,
  tools/bf16x3_numerics.hip) - the results are fp32 results, the executed MFMA work is 3x the algorithmic FLOPs.
--precision f32: fp32 matrix cores, 3x3 conv + LIF fused over T (profiles/r1_c_pmc_final.txt).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LEVELS = [(192, 384), (96, 192), (48, 96), (24, 48), (12, 24)]   # 768x1536 / (4,8,16,32,64)
C, A, K_CLS, HD = 256, 3, 9, 1024
T_RPN, T_DET = 8, 12
BATCH, ROIS_PER_IMG = 2, 1000
PEAK_F32_MFMA_TFLOPS = 157.3                                     # MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0                                   # dense bf16 (no sparsity)
PEAK_MX_MFMA_TFLOPS = 10000.0                                    # dense fp6 / fp4 block-scaled (spec, no sparsity)


WORKLOADS = {
    "cityscapes": {"levels": LEVELS, "K": 9, "T_rpn": 8, "T_det": 12, "batch": 2, "spike_rates": False,
                   "name": "cityscapes_1024x2048_b2_heads"},
    # BDD 720x1280 -> 768x1376 canvas (SURVEY.md §8 shape table), 4 images per GPU
    "bdd": {"levels": [(192, 344), (96, 172), (48, 86), (24, 43), (12, 22)], "K": 11, "T_rpn": 8, "T_det": 12, "batch": 4,
            "spike_rates": False, "name": "bdd_720x1280_b4_heads"},
    "stress": {"levels": LEVELS, "K": 9, "T_rpn": 16, "T_det": 24, "batch": 2, "spike_rates": True,
               "name": "cityscapes_1024x2048_b2_heads_T16_T24_spike_rates"},
}


def algorithmic_flops():
    pos = BATCH * sum(h * w for h, w in LEVELS)
    conv = pos * 2 * 9 * C * C * T_RPN                            # dominant kernel
    rpn = pos * 2 * (9 * C * C + C * A + C * 4 * A) * T_RPN
    det = BATCH * ROIS_PER_IMG * 2 * (C * 49 * HD + HD * HD + HD * 5 * K_CLS) * T_DET
    return conv, rpn, det


def make_inputs(dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    feats = [torch.randn((BATCH, C, h, w), generator=g).to(dev) for h, w in LEVELS]
    rois = torch.randn((BATCH * ROIS_PER_IMG, C, 7, 7), generator=g).to(dev)
    return feats, rois


def cpu_baseline():
    """the oracle (CPU restatement of the reference loops, un-fused torch ops) on a bounded sample:
    ONE of the two images — RPN head on the b=1 pyramid + detector head on its 1000 RoIs"""
    from oracle import snn_oracle as OR
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn((1, C, h, w), generator=g) for h, w in LEVELS]
    rois = torch.randn((ROIS_PER_IMG, C, 7, 7), generator=g)
    w_s = torch.randn((C, C, 3, 3), generator=g) * 0.01
    w_c = torch.randn((A, C, 1, 1), generator=g) * 0.01
    w_b = torch.randn((4 * A, C, 1, 1), generator=g) * 0.01
    w6 = (torch.rand((HD, C * 49), generator=g) * 2 - 1) / (C * 49) ** 0.5
    w7 = (torch.rand((HD, HD), generator=g) * 2 - 1) / HD ** 0.5
    wc = (torch.rand((K_CLS, HD), generator=g) * 2 - 1) / HD ** 0.5
    wb = (torch.rand((4 * K_CLS, HD), generator=g) * 2 - 1) / HD ** 0.5
    threads = torch.get_num_threads()
    with torch.no_grad():
        t0 = time.perf_counter()
        OR.rpn_head_forward(feats, w_s, w_c, w_b, T_RPN)
        OR.det_head_forward(rois, w6, w7, wc, wb, T_DET)
        dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "1 image: oracle RPN head (b=1 pyramid, T=8) + detector head (1000 RoIs, T=12), %.1f s" % dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra timing leg with the other precision")
    ap.add_argument("--precision", choices=["bf16x3", "f32", "mxfp6"], default="bf16x3")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cityscapes",
                    help="cityscapes = BASELINE.json's headline configuration (default); bdd = config[3] per-rank share "
                         "(720x1280, 4 images per GPU, K=11); stress = config[4] (T=16/24, spike-rate outputs on)")
    args = ap.parse_args()
    global LEVELS, K_CLS, T_RPN, T_DET, BATCH
    wl = WORKLOADS[args.workload]
    LEVELS, K_CLS, T_RPN, T_DET, BATCH = wl["levels"], wl["K"], wl["T_rpn"], wl["T_det"], wl["batch"]

    import snn_automotive_object_detection_amd as S
    from snn_automotive_object_detection_amd import dp, ops
    import torch.distributed as dist

    rank, local, world = dp.init_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    torch.manual_seed(1234)                                      # same weights on every rank
    rpn_head = S.RPNHeadSNN(C, A, T_RPN).to(dev)
    det_head = S.FastRCNNPredictorSNNFull(C * 49, HD, K_CLS, T_DET).to(dev)
    rpn_head.precision = det_head.precision = args.precision
    rpn_head.spike_rates = det_head.spike_rates = wl["spike_rates"]
    feats, rois = make_inputs(dev, 1000 + rank)                  # inputs resident in HBM

    def step():
        rpn_out = rpn_head(feats)
        det_out = det_head(rois)
        if wl["spike_rates"]:            # the spike-rate variants return rate tensors only (faster_rcnn.py:520-618)
            return rpn_out, det_out
        cls, deltas = det_out
        payload, counts = ops.det_exchange_payload(cls, deltas, BATCH, 100)     # top-100 RoIs per image, one launch
        return dp.all_gather_detection_tensors(payload, counts)  # no-op at world == 1

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms = dt / args.steps * 1e3
    value = world * BATCH * args.steps / dt

    # ---- the same K steps with the other matrix path of the two big contractions (outside the headline number):
    # "mxfp6" = fp4 x fp6 block-scaled MFMA on 6 digit planes per weight (passes the same parity tests)
    alt = None
    if not args.no_alt and world == 1:
        alt_prec = "bf16x3" if args.precision == "mxfp6" else "mxfp6"
        rpn_head.precision = det_head.precision = alt_prec
        for _ in range(max(1, args.warmup)):
            step()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt_alt = time.perf_counter() - t1
        alt = {"precision": alt_prec, "value": round(BATCH * args.steps / dt_alt, 3), "unit": "images/s",
               "ms_per_step": round(dt_alt / args.steps * 1e3, 4)}
        rpn_head.precision = det_head.precision = args.precision

    # ---- per-kernel timing with HIP events on the launch stream (outside the timed region) ----
    def time_ms(fn, iters):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / iters

    p = rpn_head._params()
    w_sh = rpn_head._packed_shared()
    w_hd = rpn_head._cache_heads.val
    iters = max(3, min(args.steps, 10))
    conv_ms = time_ms(lambda: ops.rpn_head_forward(feats, C, A, T_RPN, p, w_sh, w_hd, stage_mask=2), iters)
    enc_ms = time_ms(lambda: ops.rpn_head_forward(feats, C, A, T_RPN, p, w_sh, w_hd, stage_mask=1), iters)
    rpn_ms = time_ms(lambda: rpn_head(feats), iters)
    det_ms = time_ms(lambda: det_head(rois), iters)
    conv_fl, rpn_fl, det_fl = algorithmic_flops()
    P = BATCH * sum(h * w for h, w in LEVELS)
    if args.precision == "f32":
        kernel, peak, exec_factor, kernel_ms = "k_conv3x3_lif<false>", PEAK_F32_MFMA_TFLOPS, 1.0, conv_ms
        traffic_key = "f32"
    elif args.precision == "mxfp6":
        # stage 2 = k_gemm_mx<3, 4> (G3_CONV_LIF_TILE): fp4 x fp6 block-scaled MFMA, 6 digit planes per weight at 4x the K per
        # instruction: executed work = 6/4 of the bf16-equivalent; peak = 10 PF dense fp6/fp4 (spec)
        kernel, peak, exec_factor, kernel_ms = "k_gemm_mx<3, 4>", PEAK_MX_MFMA_TFLOPS, 6.0, conv_ms
        traffic_key = "mxfp6"
    else:
        # stage 2 = k_gemm_bf16x3<3, 3, 4, 2> (MODE = G3_CONV_LIF_TILE, 3-slot ring, 4 M-tiles per wave, 4 x 2 wave grid): 3x3 conv + LIF over T fused in the tile,
        # on the bf16 matrix cores
        kernel, peak, exec_factor, kernel_ms = "k_gemm_bf16x3<3, 3, 4, 2>", PEAK_BF16_MFMA_TFLOPS, 3.0, conv_ms
        traffic_key = "bf16x3"
    achieved = conv_fl / (kernel_ms * 1e-3) / 1e12             # ALGORITHMIC (dense-equivalent) TFLOP/s
    traffic = None                    # HBM bytes per launch of the dominant kernel, from the committed PMC passes
    try:
        with open(os.path.join(ROOT, "profiles", "r1_traffic.json")) as f:
            traffic = json.load(f)[traffic_key]["hbm_bytes_per_launch"]
    except Exception:
        pass

    out = {
        "metric": "images/sec (T_rpn=%d,T_det=%d, %s b=%d) spiking RPN+RoI heads forward" % (
            T_RPN, T_DET, "720x1280" if args.workload == "bdd" else "1024x2048", BATCH),
        "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"f32": "f32", "bf16x3": "f32 (weights as exact bf16x3 split, fp32 accumulate)",
                                        "mxfp6": "f32 (weights as 6 fp6 digit planes with block scales, fp32 accumulate)"}[args.precision],
        "data": "synthetic",
        "config": {"precision": args.precision,
                   "workload": "%s: RPNHeadSNN(T=%d) on 5-level pyramid %dx256x{%dx%d..%dx%d} + FastRCNNPredictorSNNFull(T=%d) on "
                               "%d RoIs x 12544, K=%d; random-init weights" % (
                                   wl["name"], T_RPN, BATCH, LEVELS[0][0], LEVELS[0][1], LEVELS[-1][0], LEVELS[-1][1], T_DET,
                                   BATCH * ROIS_PER_IMG, K_CLS),
                   "global_batch": BATCH * world, "parallelism": "dp%d" % world,
                   "exchange": "all-gather of per-image detections [100x6] (RCCL)" if world > 1 else "none"},
        # achieved = ALGORITHMIC FLOPs of the launch / its duration.  For bf16x3 the peak is what the bf16 matrix pipe
        # can deliver of this arithmetic: the dense bf16 MFMA peak / 3 MFMAs per exact fp32 product (the executed
        # rate against the full 2.5 PF is the same fraction; both are spelled out).
        "roofline": {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2),
                     "peak": round(peak / exec_factor, 1), "unit": "TFLOP/s", "frac": round(achieved * exec_factor / peak, 4),
                     "traffic": traffic, "launch_ms": round(kernel_ms, 4),
                     "algorithmic_gflop_per_launch": round(conv_fl / 1e9, 1), "executed_over_algorithmic": exec_factor,
                     "executed_tflops": round(achieved * exec_factor, 2), "mfma_peak_tflops": peak,
                     "algorithmic_frac_of_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4)},
        "breakdown_ms": {"rpn_head": round(rpn_ms, 3), "rpn_encode": round(enc_ms, 3), "rpn_conv3x3_lif": round(conv_ms, 3),
                         "det_head": round(det_ms, 3)},
        "heads_tflops": round((rpn_fl + det_fl) / ((rpn_ms + det_ms) * 1e-3) / 1e12, 2),
        "alt_precision": alt,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
