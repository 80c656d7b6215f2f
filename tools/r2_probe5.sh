#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/parity_r2.jsonl
python -m pytest tests -q -m gpu > gpurun_out/r2_t_all2.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/r2_t_all2.log | tail -20
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r2_bench3.log 2> gpurun_out/r2_bench3.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r2_bench3.log"):
    if l.startswith("{"):
        o = json.loads(l)
        print("value", o["value"], "ms", o["ms_per_step"], "roofline", o["roofline"]["launch_ms"], o["roofline"]["frac"], "breakdown", o["breakdown_ms"])
        e = o["extra"]
        print("sustained", e["sustained"]["value"], "mx", e["alt_precision"]["value"], "bdd", e["bdd"]["value"], e["bdd"]["roofline"]["frac"], "stress", e["stress"]["value"], e["stress"]["breakdown_ms"], e["stress"]["kernels_over_step"], "e2e", e["e2e"]["value"], e["e2e"]["stage_ms"])
PY
