"""Per-kernel GPU time of the end-to-end forward from a `rocprofv3 --kernel-trace --stats` run of tools/prof_e2e.py N
(tools/r4_final.sh): python tools/e2e_kernel_table.py <dir with *kernel_stats.csv> <batches> > profiles/<round>_e2e_kernels.txt"""
import csv
import glob
import os
import sys

d, n = sys.argv[1], float(sys.argv[2])
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))]
ours = [r for r in rows if r[0].replace("void ", "").startswith("k_")]
rest = [r for r in rows if r not in ours]
tot = sum(r[2] for r in rows) / n / 1e6
print("End-to-end forward (create_model('cityscapes', 9), 2 x rand(3,1024,2048), T = 8 / 12) under rocprofv3 --kernel-trace --stats, "
      "%d batches after warm-up (tools/prof_e2e.py)" % n)
print("GPU time per batch: %.2f ms" % tot)
for title, grp in (("this package's kernels", ours), ("stock backbone, transform and glue (MIOpen / rocBLAS / ATen)", rest)):
    print("%s: %.2f ms per batch" % (title, sum(r[2] for r in grp) / n / 1e6))
    for name, calls, ns in sorted(grp, key=lambda r: -r[2])[:40 if grp is ours else 12]:
        print("  %-72s calls/batch %5.1f  avg %8.1f us  per batch %8.1f us" % (name[:72], calls / n, ns / calls / 1e3, ns / n / 1e3))
