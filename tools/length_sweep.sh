#!/bin/bash
# Run ON THE GPU BOX: config-2 / config-3 shaped batches (10M reads) at the common Illumina read lengths.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out; : > gpurun_out/lengths.log
# (cfg3_150: reads with adapters in the layout the host feed gives them — stride rounded up to 4 when the length is not a multiple of 4;
#  trimmed_adapters: 70 % of the reads of that length, the rest down to 80 % of it, strided with 0xFF pads, adapter table loaded)
for w in cfg2 cfg3_150 trimmed_adapters; do
  for L in 36 50 76 100 125 150 200 250 300; do
    python bench.py --workload $w --read-len $L --no-also --no-cpu-baseline --no-tiers --no-traffic --no-steady --steps 100 --warmup 30 2>/dev/null | tail -1 > /tmp/ls.json
    python - "$w" "$L" <<'PY'
import json, sys
d = json.load(open("/tmp/ls.json")); r = d["roofline"]
name = {"cfg3_150": "cfg3 (adapters)", "trimmed_adapters": "trimmed + adapters"}.get(sys.argv[1], sys.argv[1])
line = "%-18s L=%-3s  %.3f Tbases/s  step %.4f ms  kernel %.4f ms  frac %.3f" % (name, sys.argv[2], d["value"] / 1e12, d["ms_per_step"], r["kernel_ms"], r["frac"])
print(line)
open("gpurun_out/lengths.log", "a").write(line + "\n")
PY
  done
done
