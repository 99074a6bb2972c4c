#!/bin/bash
# developer helper: one bench.py workload under several environment settings, on one box
#   tools/env_bench.sh <workload> "<bench args>" "VAR=.. VAR=.." "VAR=.." ...   ("-" = no variables)
w=$1; args=$2; shift 2
for rep in 1 2; do
  for e in "$@"; do
    [ "$e" = "-" ] && ee="" || ee="$e"
    env $ee python bench.py --workload $w --no-also --no-cpu-baseline --no-tiers --no-traffic --no-steady $args 2>/tmp/env_err.txt | tail -1 > /tmp/env_line.json
    python - "$e" "$w" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/env_line.json"))
    print("%-40s %s step %.4f ms  kernel %.4f ms  frac %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["roofline"].get("kernel_ms") or 0, d["roofline"]["frac"]))
except Exception as ex:
    print(sys.argv[1], "FAILED", open("/tmp/env_err.txt").read()[-300:])
PY
  done
done
