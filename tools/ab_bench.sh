#!/bin/bash
# A/B two builds of libquack_hip.so on one box: tools/ab_bench.sh <other.so> <workload>...
# (alternates A B A B so that clock drift hits both)
set -e
other=$1; shift
cp quack_amd/libquack_hip.so /tmp/ab_A.so
# whatever happens (a failing B run under set -e, Ctrl-C): the product library is put back
trap 'cp /tmp/ab_A.so quack_amd/libquack_hip.so' EXIT
cp "$other" /tmp/ab_B.so
for rep in $(seq 1 ${AB_REPS:-2}); do
  for v in A B; do
    cp /tmp/ab_$v.so quack_amd/libquack_hip.so
    for w in "$@"; do
      python bench.py --workload $w --no-also --no-cpu-baseline --no-tiers --no-traffic --no-steady $AB_ARGS 2>/dev/null | tail -1 > /tmp/ab_line.json
      python - "$v" "$w" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
print(sys.argv[1], sys.argv[2], "step %.4f ms  kernel %.4f ms  frac %.3f" % (d["ms_per_step"], d["roofline"].get("kernel_ms") or 0, d["roofline"]["frac"]))
PY
    done
  done
done
