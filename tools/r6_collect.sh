#!/bin/bash
# copy the evidence of `tools/r6_final.sh <tag>` (merged back into gpurun_out/) into profiles/ under its round-6 names.   usage: bash tools/r6_collect.sh <tag>   (HERE, after the lease)
TAG=${1:-r6z}
set -e
for w in default stress bdd; do cp gpurun_out/prof_${TAG}_$w/summary.txt profiles/r6_${w}_summary.txt; cp gpurun_out/prof_${TAG}_$w/kernel_stats.csv profiles/r6_${w}_kernel_stats.csv; done
cp gpurun_out/${TAG}_traffic.json profiles/r6_traffic.json
cp gpurun_out/${TAG}_bench.json profiles/r6_bench_default.json
cp gpurun_out/${TAG}_bench_driver_style.json profiles/r6_bench_driver_style.json
cp gpurun_out/${TAG}_bench_8ranks_gloo_1gpu.json profiles/r6_bench_8ranks_gloo_1gpu.json
cp gpurun_out/${TAG}_ab_knobs.txt profiles/r6_ab_knobs.txt; cp gpurun_out/${TAG}_ab_knobs_stress.txt profiles/r6_ab_knobs_stress.txt
cp gpurun_out/${TAG}_e2e_kernels.txt profiles/r6_e2e_kernels.txt; cp gpurun_out/${TAG}_timelines.txt profiles/r6_timelines.txt
DIG=$(python -c "import json; print(json.load(open('gpurun_out/${TAG}_traffic.json'))['source_digest'][:12])")
{ echo "# round 6, final tree (source digest $DIG), one lease (tools/r6_final.sh $TAG): python -m pytest tests -q -m gpu --durations=25"; tail -40 gpurun_out/${TAG}_tests.log; } > profiles/r6_gpu_pytest.txt
{ echo "# round 6, final tree (source digest $DIG), same lease: python -m pytest tests -q -m \"gpu and sweep\" --durations=10   (the sweep-marked tests)"; tail -25 gpurun_out/${TAG}_sweep_tests.log; } > profiles/r6_sweep_pytest.txt
{ echo "# round 6, final tree (source digest $DIG), same lease: SNN_TEST_SWEEP=1 python -m pytest tests -q -m gpu --durations=15   = EVERY GPU test with its full grid (the sampled loops of the"; echo "# unmarked tests at every T = 2 .. 26 / every shape, and the sweep-marked tests); 8 skipped = the ping-pong conv's tests, which need a -DSNN_PINGPONG build"; tail -24 gpurun_out/${TAG}_exhaustive_tests.log; } > profiles/r6_exhaustive_pytest.txt
cp gpurun_out/${TAG}_parity_watch.txt profiles/r6_parity_watch.txt
python - <<PY
import json
recs = [json.loads(l) for l in open('gpurun_out/parity_r6.jsonl') if l.strip()]
json.dump(recs, open('profiles/parity_r6.json', 'w'), indent=1)
print(len(recs), "parity records")
PY
python -c "
from snn_automotive_object_detection_amd import build; import json
print('tree digest', build.source_digest()[:12], '| traffic json digest', json.load(open('profiles/r6_traffic.json'))['source_digest'][:12])"
