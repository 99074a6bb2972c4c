#!/bin/bash
# Run ON THE GPU BOX (via gpurun): PC sampling of one bench workload (rocprofv3, beta) -> gpurun_out/pcs_<tag>/
#   tools/pc_sample.sh <tag> <method: host_trap|stochastic> <interval> <bench args...>
# tools/pc_summary.py maps the sampled code-object offsets back to the kernel's instructions.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; METHOD=$2; IVAL=$3; shift 3
OUT=$ROOT/gpurun_out/pcs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
UNIT=time; [ "$METHOD" = stochastic ] && UNIT=cycles
timeout -k 10 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $IVAL \
  --kernel-trace --output-format csv json -d $OUT/run -- python3 $ROOT/bench.py --no-cpu-baseline --no-also --no-tiers --no-traffic --no-steady $* > $OUT/run.log 2>&1
echo "rc=$?" >> $OUT/run.log
tail -5 $OUT/run.log
find $OUT/run -type f | head -20
for f in $(find $OUT/run -name "*pc_sampling*csv" | head -3); do echo "== $f"; head -5 $f; wc -l $f; done
# keep what comes back small: per-offset counts only
python3 - "$OUT" <<'PY'
import csv, glob, collections, json, sys, os
out = sys.argv[1]
for f in glob.glob(out + "/run/**/*pc_sampling*csv", recursive=True):
    cnt = collections.Counter()
    cols = None
    with open(f) as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        for row in rd:
            key = tuple(row.get(k, "") for k in ("Code_Object_Id", "Code_Object_Offset", "Instruction", "Instruction_Comment", "Stall_Reason", "Wave_Issued", "Instruction_Type") if k in row)
            cnt[key] += 1
    with open(os.path.join(out, os.path.basename(f) + ".counts.json"), "w") as o:
        json.dump({"columns": cols, "counts": [[list(k), v] for k, v in cnt.most_common()]}, o)
    os.remove(f)
PY
ls -la $OUT
