"""Issue-side / stall counters of the big launches (round 6): what does a SIMD issue while it is NOT in a matrix instruction?
Runs on the GPU box.  The parent never touches the GPU; every pass is `rocprofv3 --pmc ... -- python3 bench.py ...` as a child (counters only
with --kernel-trace, as the pool requires), one pass per group of counters that fits the SQ's counter registers.  Counter names the box does
not list (`rocprofv3 --list-avail`) are dropped, so a pass cannot fail on a name.

usage: python3 tools/stall_counters.py <outdir> [bench args ...]      -> <outdir>/stalls.txt (+ avail.txt, raw CSVs under pass*/)
"""
import csv
import glob
import os
import re
import subprocess
import sys
from collections import defaultdict

WANT = [
    # pass: what is issued (instruction counts per category)
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"],
    ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT", "SQ_INSTS_FLAT_LDS_ONLY", "SQ_INSTS_GDS", "SQ_INSTS_BRANCH", "SQ_INSTS_SENDMSG", "SQ_INSTS_VSKIPPED"],
    # pass: cycles a wave spends executing each category (summed over waves)
    ["SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_EXP_GDS"],
    ["SQ_INST_CYCLES_SALU", "SQ_INST_CYCLES_SMEM", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_INST_CYCLES_VMEM", "SQ_THREAD_CYCLES_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"],
    # pass: what waves wait for
    ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_IFETCH", "SQ_IFETCH", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
    # pass: LDS pipe
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_UNALIGNED_STALL", "SQ_LDS_MEM_VIOLATIONS", "SQ_LDS_ATOMIC_RETURN", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL"],
    # pass: instruction levels (outstanding instructions accumulated per cycle) and issue
    ["SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_SMEM", "SQ_LEVEL_WAVES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_BF16", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU"],
    ["SQ_INSTS_WAVE32_LDS", "SQ_WAVES_EQ_64", "SQ_WAVES_LT_64", "SQ_ITEMS", "SQ_CYCLES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACCUM_PREV"],
    # (a pass of TCP_* / TA_* counters hung the profiled process for 25 minutes on this pool: not collected)
]
KERNELS = ("gemm_lif_sparse", "gemm_bf16x3", "li_heads", "encode")


def avail(outdir):
    r = subprocess.run(["rocprofv3", "--list-avail"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    with open(os.path.join(outdir, "avail.txt"), "w") as f:
        f.write(r.stdout)
    return set(re.findall(r"\b([A-Z][A-Za-z0-9]*_[A-Za-z0-9_]+)\b", r.stdout))


def main():
    outdir = os.path.abspath(sys.argv[1])
    bench_args = sys.argv[2:]
    os.makedirs(outdir, exist_ok=True)
    os.environ["TMPDIR"] = "/tmp"
    names = avail(outdir)
    lines = []
    table = defaultdict(dict)
    for i, want in enumerate(WANT):
        have = [c for c in want if c in names]
        missing = [c for c in want if c not in names]
        if missing:
            lines.append("pass %d: not on this box: %s" % (i, " ".join(missing)))
        if not have:
            continue
        d = os.path.join(outdir, "pass%d" % i)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + have + ["--output-format", "csv", "-d", d, "--",
               "python3", "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--no-clock-probe"] + bench_args
        try:                                                     # (a pass of TCP counters once sat for 25 minutes: every pass has its own limit)
            r = subprocess.run(["timeout", "-k", "10", "240"] + cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        except Exception as e:
            lines.append("pass %d: %r" % (i, e))
            continue
        with open(os.path.join(outdir, "pass%d.log" % i), "w") as f:
            f.write(" ".join(cmd) + "\n" + r.stdout[-6000:])
        if r.returncode == 124:
            lines.append("pass %d: timed out after 240 s (%s)" % (i, " ".join(have)))
        acc = defaultdict(lambda: defaultdict(list))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if not acc:
            lines.append("pass %d: rc %d, no counter rows (%s)" % (i, r.returncode, " ".join(have)))
        for k, cs in acc.items():
            if any(x in k for x in KERNELS):
                for c, v in cs.items():
                    table[k][c] = (sum(v) / len(v), len(v))
        for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True):
            if os.path.getsize(f) > (2 << 20):
                os.remove(f)
        write_summary(outdir, bench_args, lines, table)          # (after every pass: a later pass that hangs loses nothing)
    print(open(os.path.join(outdir, "stalls.txt")).read())


def write_summary(outdir, bench_args, lines, table):
    with open(os.path.join(outdir, "stalls.txt"), "w") as f:
        f.write("# mean per dispatch, `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra %s`, one rocprofv3 --pmc pass per counter group\n" % " ".join(bench_args))
        for l in lines:
            f.write("# " + l + "\n")
        for k in sorted(table):
            f.write("\n== %s\n" % k[:110])
            for c in sorted(table[k]):
                f.write("  %-34s %16.6g  (n=%d)\n" % (c, table[k][c][0], table[k][c][1]))


if __name__ == "__main__":
    main()
