#!/bin/bash
# Run ON A ONE-GPU BOX: the driver's N>1 command line with NR ranks sharing device 0 and a gloo exchange — the plumbing of
# the multi-GPU bench line (ranks.parity_check, n1_reference, also.cfg3_share) and what each phase costs in wall time, not a
# scaling number.  NR defaults to 5: the GPU boxes of this pool allow at most 6 processes on the card at once (six ranks plus the
# launcher were counted as seven and the run was killed), so the N = 8 line
# the driver will run cannot be rehearsed here with 8 ranks on one device.
#   tools/rehearse_multi.sh [NR] [bench args...]
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
NR=${1:-5}; shift
( time python -c "import torch" ) > gpurun_out/rehearse_import.log 2>&1
OUT=gpurun_out/rehearsal_gloo_${NR}ranks_one_gpu.json
T0=$(date +%s.%N)
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $NR --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $NR --backend gloo --device 0 \
  --steps 20 --warmup 5 "$@" > $OUT 2> gpurun_out/rehearse_err.log || { tail -20 gpurun_out/rehearse_err.log; exit 1; }
python - $OUT $T0 <<'PY'
import json, sys, time
d = json.load(open(sys.argv[1]))
print("wall of the whole command: %.1f s" % (time.time() - float(sys.argv[2])))
print("n_gpus", d["n_gpus"], "value %.4g" % d["value"], "parity_check", d["ranks"]["parity_check"]["ok"], d["ranks"]["parity_check"]["reads"], "efficiency", round(d["efficiency_vs_n1_reference"], 3))
print("phases (rank 0):", d.get("phases_s"), "dropped:", (d.get("budget") or {}).get("dropped"))
PY
