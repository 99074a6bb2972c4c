#!/bin/bash
# Run ON THE GPU BOX: tools/profile_round.sh for the round's workloads (progress into gpurun_out/${R}_profiles.log)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=${R:-r05}   # the round tag of the output files
cd $ROOT; mkdir -p gpurun_out
( time python -c "import torch" ) > gpurun_out/${R}_profiles_import.log 2>&1
for w in ${@:-cfg2 cfg3 cfg3_150 cfg5 trimmed trimmed_adapters}; do
  echo "== $w $(date +%T)" >> gpurun_out/${R}_profiles.log
  if [ $w = cfg2 ]; then bash tools/profile_round.sh ${R}_$w >> gpurun_out/${R}_profiles.log 2>&1 || exit 1
  else bash tools/profile_round.sh ${R}_$w --workload $w >> gpurun_out/${R}_profiles.log 2>&1 || exit 1; fi
  # keep what the summariser needs, drop the bulky rest
  find gpurun_out/prof_${R}_$w -name "*_agent_info.csv" -delete
  # (the batch is made by hundreds of small torch kernels, which the counter passes record too: keep the histogram kernel's rows)
  for f in $(find gpurun_out/prof_${R}_$w -name "*_counter_collection.csv" -size +1M); do
    { head -1 $f; grep hist_kernel $f; } > $f.tmp && mv $f.tmp $f
  done
done
du -sh gpurun_out/prof_${R}_* | tail -5
