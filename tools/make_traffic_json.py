"""Build profiles/r6_traffic.json (round 5: r5_traffic.json) from the summaries of tools/prof_round.sh (kernel stats + the five separate PMC passes), one per bench
workload:   python tools/make_traffic_json.py profiles/r5_traffic.json cityscapes=profiles/r5_default_summary.txt stress=... bdd=...
Per workload and big launch (conv = k_gemm_lif_sparse<true, ...>, fc6 = k_gemm_lif_sparse<false, ...>, fc7 = k_gemm_bf16x3<4, ...>): HBM bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE, counter unit 1 KiB: MI355X_MICROARCH.md, HBM section; calibrated for this family's LDS-DMA gathers by
tools/fetch_calib.hip), matrix-pipe busy fraction, clock, matrix instructions, L2 hits / misses - what bench.py quotes as `roofline.traffic`
of each leg and what a reader needs to recompute every leg's `frac`."""
import csv
import json
import os
import re
import sys

WORKLOADS = {  # shapes of bench.py's workloads: positions of the pyramid (batch included), padded positions, T_rpn, RoIs, T_det, spike rates
    "cityscapes": dict(levels=[(192, 384), (96, 192), (48, 96), (24, 48), (12, 24)], batch=2, T_rpn=8, R=2000, T_det=12, rates=False),
    "bdd": dict(levels=[(192, 344), (96, 172), (48, 86), (24, 43), (12, 22)], batch=4, T_rpn=8, R=4000, T_det=12, rates=False),
    "stress": dict(levels=[(192, 384), (96, 192), (48, 96), (24, 48), (12, 24)], batch=2, T_rpn=16, R=2000, T_det=24, rates=True),
}
KERNELS = {"conv": "k_gemm_lif_sparse<true,", "fc6": "k_gemm_lif_sparse<false,", "fc7": "k_gemm_bf16x3<4,",
           "enc_rpn": "k_encode_levels<", "enc_det": "k_encode_rows_perm<"}          # (the two encoders: HBM-bound streaming launches)


def parse(path):
    stats, pmc = {}, {}
    for line in open(path):
        m = re.match(r"\s+(.*?)\s+calls\s+(\d+)\s+avg\s+([\d.]+) us", line)
        if m:
            stats[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
            continue
        cs = re.findall(r"(\w+)=([\d.e+-]+) \(n=(\d+)\)", line)
        if cs:
            name = line[2:62].strip()
            pmc.setdefault(name, {}).update({c: (float(v), int(n)) for c, v, n in cs})
    # the summary lists the 14 longest kernels only (under the profiler MIOpen's trial kernels of the input-making backbone can crowd the small
    # launches out): the full kernel-stats table next to it fills in the rest
    for sib in (path.replace("_summary.txt", "_kernel_stats.csv"), os.path.join(os.path.dirname(path), "kernel_stats.csv")):
        if sib != path and os.path.exists(sib):
            for r in csv.DictReader(open(sib)):
                stats.setdefault(r["Name"][:70], (int(r["Calls"]), round(float(r["AverageNs"]) / 1e3, 1)))
            break
    return stats, pmc


def find(d, prefix):
    hits = [k for k in d if prefix in k]
    return (hits[0], d[hits[0]]) if hits else (None, None)


def operands(wl, which):
    """algorithmic HBM bytes of a launch: what it must read and write once (period planes in, weights, spike planes out)"""
    w = WORKLOADS[wl]
    P = w["batch"] * sum(h * x for h, x in w["levels"])
    Pe = w["batch"] * sum((h + 2) * (x + 2) for h, x in w["levels"])
    if which == "conv":
        Tc = w["T_rpn"] - 1
        dense = 2 * Pe * 32                                    # two raw planes: 8 words per padded position
        sparse = (Tc - 2) * 4 * 16 * Pe                        # compressed: 4 steps of 64 k x 4 dwords
        return dense + sparse + 3.5e6 + w["T_rpn"] * P * 32, "2 raw + %d compressed period planes in, 3.5 MB weight planes, %d spike planes out" % (Tc - 2, w["T_rpn"])
    if which == "fc6":
        Tc = w["T_det"] - (1 if w["rates"] else 2)
        R = w["R"]
        return 2 * R * 392 * 4 + (Tc - 2) * 196 * 16 * R + 77.1e6 + w["T_det"] * R * 128, "2 raw + %d compressed planes of %d RoIs in, 77 MB weight planes, lif6's planes out" % (Tc - 2, R)
    if which == "enc_rpn":                                     # features in (fp32), 2 raw + Tc - 2 compressed period planes out
        Tc = w["T_rpn"] - 1
        return P * 256 * 4 + 2 * Pe * 32 + (Tc - 2) * 4 * 16 * Pe, "fp32 features of %d positions x 256 channels in, 2 raw + %d compressed period planes out" % (P, Tc - 2)
    if which == "enc_det":                                     # RoI features in (fp32), planes out in fc6's order
        Tc = w["T_det"] - (1 if w["rates"] else 2)
        R = w["R"]
        return R * 12544 * 4 + 2 * R * 392 * 4 + (Tc - 2) * 196 * 16 * R, "fp32 RoI features [%d, 12544] in, 2 raw + %d compressed period planes out" % (R, Tc - 2)
    return None, ""


def main():
    out_path, pairs = sys.argv[1], [a.split("=", 1) for a in sys.argv[2:]]
    res = {"note": "built by tools/make_traffic_json.py from the tools/prof_round.sh summaries named in `source` (separate rocprofv3 --pmc passes: FETCH_SIZE, WRITE_SIZE, "
                   "SQ_* + GRBM_GUI_ACTIVE, TCC_HIT / MISS; kernel stats from --kernel-trace --stats of `bench.py --steps 20 --warmup 3 --no-extra`); "
                   "hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies the 128-B requests of wide reads at 64 B: MI355X_MICROARCH.md; calibrated for this "
                   "family's LDS-DMA gathers by tools/fetch_calib.hip); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"}
    for wl, path in pairs:
        stats, pmc = parse(path)
        entry = {"source": path}
        for key, prefix in KERNELS.items():
            sname, st = find(stats, prefix)
            pname, pc = find(pmc, prefix)
            if not st or not pc:
                continue
            e = {"kernel": sname, "dispatches_in_stats": st[0], "avg_us": st[1]}
            g = lambda c: pc.get(c, (None, 0))[0]
            if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
                e["fetch_size_counter_kib"], e["write_size_counter_kib"] = g("FETCH_SIZE"), g("WRITE_SIZE")
                e["hbm_bytes_per_launch"] = int(g("FETCH_SIZE") * 1024 * 2 + g("WRITE_SIZE") * 1024)
                e["hbm_gb_per_s"] = round(e["hbm_bytes_per_launch"] / (st[1] * 1e-6) / 1e9, 1)
                e["hbm_frac_of_8tb_s"] = round(e["hbm_gb_per_s"] / 8000.0, 4)
                e["hbm_frac_of_6p3tb_s_copy_rate"] = round(e["hbm_gb_per_s"] / 6300.0, 4)      # (what a float4 copy reaches: MI355X_MICROARCH.md)
            ob, onote = operands(wl, key)
            if ob:
                e["algorithmic_hbm_bytes"], e["algorithmic_note"] = int(ob), onote
                if "hbm_bytes_per_launch" in e:
                    e["traffic_over_operands"] = round(e["hbm_bytes_per_launch"] / ob, 2)
            if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE"):
                e["mfma_busy"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * g("GRBM_GUI_ACTIVE") / 8.0), 4)
                e["clock_ghz_profiled"] = round(g("GRBM_GUI_ACTIVE") / 8.0 / (st[1] * 1e-6) / 1e9, 3)
                e["matrix_insts_per_launch"] = g("SQ_INSTS_MFMA")
                e["valu_insts_per_launch"] = g("SQ_INSTS_VALU")
                e["lds_bank_conflict_cycles"] = g("SQ_LDS_BANK_CONFLICT")
            if g("TCC_HIT_sum") is not None:
                e["l2_hits"], e["l2_misses"] = g("TCC_HIT_sum"), g("TCC_MISS_sum")
            entry[key] = e
        res[wl] = {"bf16x3": entry}
    # the source tree the PMC passes were taken on: bench.py quotes `roofline.traffic` from this file only for a build of the SAME digest
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from snn_automotive_object_detection_amd import build as _build
    res["source_digest"] = _build.source_digest()
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main()
