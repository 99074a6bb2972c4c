#!/bin/bash
# Run ON THE GPU BOX: the product library against several experimental builds, in turn, REPS rounds:
#   tools/ab_many.sh <workload> <lib.so>... -- [bench args]
w=$1; shift
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ "$1" = "--" ] && shift
cp quack_amd/libquack_hip.so /tmp/ab_A.so
trap 'cp /tmp/ab_A.so quack_amd/libquack_hip.so' EXIT
for rep in $(seq 1 ${REPS:-2}); do
  for l in /tmp/ab_A.so "${libs[@]}"; do
    [ "$l" = /tmp/ab_A.so ] || cp "$l" /tmp/ab_X.so
    [ "$l" = /tmp/ab_A.so ] && cp /tmp/ab_A.so quack_amd/libquack_hip.so || cp /tmp/ab_X.so quack_amd/libquack_hip.so
    python bench.py --workload $w --no-also --no-cpu-baseline --no-tiers --no-traffic --no-steady "$@" 2>/tmp/ab_err.txt | tail -1 > /tmp/ab_line.json
    python - "$(basename $l)" "$w" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/ab_line.json"))
    r = d["roofline"]
    print("%-24s %-8s step %.4f ms  kernel %.4f ms (%.4f..%.4f)  frac %.3f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_max"], r["frac"]))
except Exception:
    print(sys.argv[1], "FAILED", open("/tmp/ab_err.txt").read()[-300:])
PY
  done
done
