#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/parity_r2.jsonl
python -m pytest tests -q -m gpu > gpurun_out/r2_t_all3.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|FAILED|Error" gpurun_out/r2_t_all3.log | tail -10
timeout 900 python bench.py > gpurun_out/r2_bench4.log 2> gpurun_out/r2_bench4.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r2_bench4.log"):
    if l.startswith("{"):
        o = json.loads(l)
        print("value", o["value"], "ms", o["ms_per_step"], "roofline", o["roofline"]["launch_ms"], o["roofline"]["frac"], "traffic", o["roofline"]["traffic"], "breakdown", o["breakdown_ms"])
        e = o["extra"]
        print("sustained", e["sustained"]["value"], "mx", e["alt_precision"]["value"], "bdd", e["bdd"]["value"], e["bdd"]["roofline"]["frac"], "stress", e["stress"]["value"], e["stress"]["kernels_over_step"], "e2e", e["e2e"]["value"], e["e2e"]["stage_ms"], e["e2e"]["detections"])
        print("cpu", o["cpu_baseline"])
PY
