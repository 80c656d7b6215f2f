# per-kernel times of the detector head on the bench's RoI features (rocprofv3 --kernel-trace --stats over tools/det_pair_probe.py run 20)
export TMPDIR=/tmp
rm -rf gpurun_out/fc6p
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fc6p -- python3 tools/det_pair_probe.py run 20 > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/fc6p/*/*_kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f))):
    if r["Name"].startswith(("void k_", "k_")) and int(r["Calls"]) >= 20:
        print("%-84s calls %5s avg %9.1f us  min %9.1f" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"])/1e3))
PY
