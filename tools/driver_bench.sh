#!/bin/bash
# the driver's bench command (N=1), timed; line -> gpurun_out/<tag>.json
set -o pipefail
mkdir -p gpurun_out
TAG=${1:-r5_bench}; shift
( time python -c "import torch" ) > gpurun_out/${TAG}_import.log 2>&1
T0=$(date +%s.%N); python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > gpurun_out/${TAG}.json 2> gpurun_out/${TAG}_err.log; rc=$?
echo "bench wall: $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $T0) s"
[ $rc -eq 0 ] || { tail -30 gpurun_out/${TAG}_err.log; exit $rc; }
python - $TAG <<'PY'
import json, sys
d = json.load(open("gpurun_out/%s.json" % sys.argv[1]))
r = d["roofline"]
print("value %.4g ms/step %.4f frac %.4f first_window %s steady %s traffic %s" % (d["value"], d["ms_per_step"], r["frac"], (r.get("first_window") or {}).get("frac"), (r.get("steady_state") or {}).get("frac"), r.get("traffic_over_algorithmic")))
print("  phases:", d.get("phases_s"), "dropped:", (d.get("budget") or {}).get("dropped"))
for k, v in (d.get("also") or {}).items():
    rr = v["roofline"]
    print("  also.%s: frac %.4f whole %.4f kernel_ms %.4f traffic %s cpu %s" % (k, rr["frac"], rr["frac_whole_batch"], rr["kernel_ms"], rr.get("traffic_over_algorithmic"), (v.get("cpu_baseline") or {}).get("value")))
t = d.get("tiers") or {}
print("  h2d:", (t.get("h2d_inclusive") or {}).get("value"))
e = t.get("end_to_end") or {}
for k in ("config2", "config2_adapters", "config3", "config5", "paired", "single_member"):
    x = e.get(k) or {}
    print("  e2e.%s:" % k, x.get("Gbases_per_s"), x.get("wall_s"), x.get("counters_whole_file"), (x.get("counters_first_member") or {}).get("ok"), (x.get("cpu_baseline") or {}).get("value"), x.get("skipped"), x.get("wall_over_16_member_wall"))
print("  sustained:", e.get("sustained"))
print("  cpu_baseline:", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline_threads") or {}).get("value"))
PY
