"""Tie-flip drift over the kept parity records (CPU): for every record of profiles/parity_r*.json (and gpurun_out/parity_r6.jsonl if
present) that carries a budget and an observed count, observed / expectation (the budget inverted: budget = lambda + 4 sqrt(lambda) + 2) and
whether it sits above lambda + 3 sqrt(lambda).  A test that is above 3 sigma in two consecutive records is printed as a REPEAT offender and
the exit code is 1.   usage: python tools/parity_watch.py"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
recs = []
for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "parity_r*.json"))):
    recs += [(os.path.basename(f), r) for r in json.load(open(f))]
jl = os.path.join(ROOT, "gpurun_out", "parity_r6.jsonl")
if os.path.exists(jl):
    recs += [("gpurun_out/parity_r6.jsonl", json.loads(l)) for l in open(jl) if l.strip()]
last, repeat = {}, []
for src, r in recs:
    obs = next((r[k] for k in ("positions_off_tolerance", "rois_off_tolerance") if k in r), None)
    if obs is None or "budget" not in r:
        continue
    x = -2.0 + (2.0 + float(r["budget"])) ** 0.5
    lam = x * x
    over = obs > lam + 3.0 * x
    key = (r["test"], r.get("precision"), r.get("full"), r.get("fused_roialign"))
    print("%-28s %-40s observed %4d  expected %6.1f  ratio %5.2f  budget %6.1f %s" % (src, "/".join(str(k) for k in key if k is not None), obs, lam, obs / lam if lam else 0.0,
                                                                                     r["budget"], "  > 3 sigma" if over else ""))
    if over and last.get(key):
        repeat.append(key)
    last[key] = over
for k in repeat:
    print("REPEAT offender (above 3 sigma twice in a row):", k)
sys.exit(1 if repeat else 0)
