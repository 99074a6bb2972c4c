#!/bin/bash
# round 4: quick GPU check — selected tests, then A/B bench lines (args: test filter)
set -o pipefail
mkdir -p gpurun_out
( time python -c "import torch; print(torch.__version__)" ) > gpurun_out/r4_quick_import.log 2>&1   # (a fresh box pages the image in: minutes, once)
B="--steps 50 --warmup 100 --no-also --no-cpu-baseline --no-tiers --no-traffic --no-steady"
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -v -x -k "${1:-padded or grouped}" > gpurun_out/r4_quick_tests.log 2>&1 || { tail -40 gpurun_out/r4_quick_tests.log; exit 1; }
tail -3 gpurun_out/r4_quick_tests.log
rm -f gpurun_out/r4_quick_bench.log
for w in ${2:-cfg3_150 cfg3_150packed cfg3}; do
  for e in "" ${3:-QUACK_HIP_NO_GROUP=1}; do
  echo "== $w $e" >> gpurun_out/r4_quick_bench.log
  env $e timeout -k 10 300 python bench.py --workload $w $B $4 >> gpurun_out/r4_quick_bench.log 2>gpurun_out/r4_quick_err.log || { tail -20 gpurun_out/r4_quick_err.log; exit 1; }
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r4_quick_bench.log"):
    if l.startswith("=="): name=l.strip(); continue
    d=json.loads(l); r=d["roofline"]
    print(name, "ms/step %.4f kernel %.4f frac %.4f" % (d["ms_per_step"], r["kernel_ms"], r["frac"]))
PY
