#!/bin/bash
# Run ON THE GPU BOX: several environments of one build in turn, REPS rounds (default 4):  tools/env_abc.sh <workload> "<env A>" "<env B>" ... -- [bench args]
w=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" = "--" ] && shift
for rep in $(seq 1 ${REPS:-4}); do
  for e in "${envs[@]}"; do
    [ "$e" = "-" ] && ee="" || ee="$e"
    env $ee python bench.py --workload $w --no-also --no-cpu-baseline --no-tiers --no-traffic --no-steady "$@" 2>/tmp/ab_err.txt | tail -1 > /tmp/ab_line.json
    python - "${ee:-(default)}" "$w" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/ab_line.json"))
    r = d["roofline"]
    print("%-52s %-8s step %.4f ms  kernel %.4f ms (%.4f..%.4f)  frac %.3f  whole %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_max"], r["frac"], r["frac_whole_batch"]))
except Exception:
    print(sys.argv[1], "FAILED", open("/tmp/ab_err.txt").read()[-400:])
PY
  done
done
