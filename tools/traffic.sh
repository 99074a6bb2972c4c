#!/bin/bash
# traffic (PMC child passes) + kernel time of workloads under env settings: tools/r4_traffic.sh <tag> 'workload|ENV=..' ...
set -o pipefail
mkdir -p gpurun_out
TAG=$1; shift
( time python -c "import torch" ) > gpurun_out/${TAG}_import.log 2>&1
B="--steps 50 --warmup 100 --no-also --no-cpu-baseline --no-tiers --no-steady"
rm -f gpurun_out/${TAG}.log
for spec in "$@"; do
  IFS='|' read -r w e <<< "$spec"
  echo "== $w $e" >> gpurun_out/${TAG}.log
  env $e timeout -k 10 400 python bench.py --workload $w $B >> gpurun_out/${TAG}.log 2>gpurun_out/${TAG}_err.log || { tail -20 gpurun_out/${TAG}_err.log; exit 1; }
done
python - $TAG <<'PY'
import json, sys
for l in open("gpurun_out/%s.log" % sys.argv[1]):
    if l.startswith("=="): name = l.strip(); continue
    d = json.loads(l); r = d["roofline"]
    print(name, "step %.4f kernel %.4f frac %.4f traffic/alg %s" % (d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("traffic_over_algorithmic")))
PY
