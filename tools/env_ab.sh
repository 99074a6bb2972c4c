#!/bin/bash
# Run ON THE GPU BOX: one build, two environments, alternating (A B A B) so that clock drift hits both:
#   tools/env_ab.sh "<VAR=.. for B>" <workload> [bench args...]       (A = no extra variables)
benv=$1; w=$2; shift 2
for rep in 1 2; do
  for v in A B; do
    [ $v = B ] && ee="$benv" || ee=""
    env $ee python bench.py --workload $w --no-also --no-cpu-baseline --no-tiers --no-traffic --no-steady "$@" 2>/tmp/ab_err.txt | tail -1 > /tmp/ab_line.json
    python - "$v ${ee:-(default)}" "$w" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/ab_line.json"))
    r = d["roofline"]
    print("%-40s %-8s step %.4f ms  kernel %.4f ms (%.4f..%.4f)  frac %.3f  whole %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_max"], r["frac"], r["frac_whole_batch"]))
except Exception:
    print(sys.argv[1], "FAILED", open("/tmp/ab_err.txt").read()[-400:])
PY
  done
done
