#!/bin/bash
# Run ON THE GPU BOX: per-launch fixed cost of the histogram kernels at a short read length, by a straight-line fit of the
# kernel time over the number of reads per launch (2.5M .. 20M reads of L bp): time = fixed + per_read x reads.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out; : > gpurun_out/fixed_cost.log
L=${1:-36}
for w in "cfg2" "cfg3_150" "cfg3_150 --splice 0"; do
  : > /tmp/fc.txt
  for n in 2500000 5000000 10000000 20000000; do
    python bench.py --workload $w --read-len $L --reads $n --no-also --no-cpu-baseline --no-tiers --no-traffic --no-steady --steps 100 --warmup 30 2>/dev/null | tail -1 > /tmp/ls.json
    python -c "import json; d=json.load(open('/tmp/ls.json')); print($n, d['roofline']['kernel_ms'], d['ms_per_step'])" >> /tmp/fc.txt
  done
  python - "$w" "$L" <<'PY' | tee -a gpurun_out/fixed_cost.log
import sys, numpy as np
rows = np.array([[float(x) for x in l.split()] for l in open("/tmp/fc.txt")])
b, a = np.polyfit(rows[:, 0], rows[:, 1], 1)
print("%-22s L=%s  kernel ms at 2.5/5/10/20M reads: %s  -> fixed %.1f us + %.2f us per million reads (step - kernel: %s us)" % (
    sys.argv[1], sys.argv[2], " ".join("%.4f" % v for v in rows[:, 1]), a * 1e3, b * 1e9, " ".join("%.1f" % ((s - k) * 1e3) for k, s in rows[:, 1:3])))
PY
done
