#!/bin/bash
# round 4: bench lines for a list of "workload|ENV=..|extra bench args" triples, alternated `reps` times on one box
#   tools/r4_ab.sh <tag> <reps> 'cfg3_150||' 'cfg3_150|QUACK_HIP_NO_GROUP=1|--splice 0' ...
set -o pipefail
mkdir -p gpurun_out
TAG=$1; REPS=$2; shift 2
( time python -c "import torch" ) > gpurun_out/${TAG}_import.log 2>&1
B="--steps 50 --warmup 100 --no-also --no-cpu-baseline --no-tiers --no-traffic --no-steady"
rm -f gpurun_out/${TAG}.log
for rep in $(seq $REPS); do
  for spec in "$@"; do
    IFS='|' read -r w e x <<< "$spec"
    echo "== $w $e $x" >> gpurun_out/${TAG}.log
    env $e timeout -k 10 300 python bench.py --workload $w $B $x >> gpurun_out/${TAG}.log 2>gpurun_out/${TAG}_err.log || { tail -20 gpurun_out/${TAG}_err.log; exit 1; }
  done
done
python - $TAG <<'PY'
import json, sys, collections
acc = collections.OrderedDict()
for l in open("gpurun_out/%s.log" % sys.argv[1]):
    if l.startswith("=="): name = l.strip(); continue
    d = json.loads(l); r = d["roofline"]
    acc.setdefault(name, []).append((d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("batch_ms")))
for k, v in acc.items():
    print(k, " | ".join("step %.4f kernel %.4f frac %.4f" % x[:3] for x in v))
PY
