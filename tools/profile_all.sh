#!/bin/bash
# Run ON THE GPU BOX: tools/profile_round.sh for the round's workloads (progress into gpurun_out/r4_profiles.log)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT; mkdir -p gpurun_out
( time python -c "import torch" ) > gpurun_out/r4_profiles_import.log 2>&1
for w in ${@:-cfg2 cfg3 cfg3_150 cfg5 trimmed}; do
  echo "== $w $(date +%T)" >> gpurun_out/r4_profiles.log
  if [ $w = cfg2 ]; then bash tools/profile_round.sh r04_$w >> gpurun_out/r4_profiles.log 2>&1 || exit 1
  else bash tools/profile_round.sh r04_$w --workload $w >> gpurun_out/r4_profiles.log 2>&1 || exit 1; fi
  # keep what the summariser needs, drop the bulky rest
  find gpurun_out/prof_r04_$w -name "*_agent_info.csv" -delete
  # (the batch is made by hundreds of small torch kernels, which the counter passes record too: keep the histogram kernel's rows)
  for f in $(find gpurun_out/prof_r04_$w -name "*_counter_collection.csv" -size +1M); do
    { head -1 $f; grep hist_kernel $f; } > $f.tmp && mv $f.tmp $f
  done
done
du -sh gpurun_out/prof_r04_* | tail -5
