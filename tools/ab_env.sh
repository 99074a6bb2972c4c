#!/bin/bash
# like ab_bench.sh, but the B build also gets environment variables: tools/ab_env.sh <other.so> "<VAR=..>" <workload> [bench args]
other=$1; benv=$2; w=$3; shift 3
cp quack_amd/libquack_hip.so /tmp/ab_A.so
trap 'cp /tmp/ab_A.so quack_amd/libquack_hip.so' EXIT   # also on Ctrl-C or a failing run
cp "$other" /tmp/ab_B.so
for rep in 1 2; do
  for v in A B; do
    cp /tmp/ab_$v.so quack_amd/libquack_hip.so
    [ $v = B ] && ee="$benv" || ee=""
    env $ee python bench.py --workload $w --no-also --no-cpu-baseline --no-tiers --no-traffic --no-steady "$@" 2>/tmp/ab_err.txt | tail -1 > /tmp/ab_line.json
    python - "$v $ee" "$w" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/ab_line.json"))
    print(sys.argv[1], sys.argv[2], "step %.4f ms  kernel %.4f ms  frac %.3f" % (d["ms_per_step"], d["roofline"].get("kernel_ms") or 0, d["roofline"]["frac"]))
except Exception:
    print(sys.argv[1], "FAILED", open("/tmp/ab_err.txt").read()[-400:])
PY
  done
done
