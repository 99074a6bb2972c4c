#!/bin/bash
# Run ON A ONE-GPU BOX: the driver's N>1 command line with two ranks sharing device 0 and a gloo exchange — the plumbing of
# the multi-GPU bench line (ranks.parity_check, n1_reference, also.cfg3_share), not a scaling number
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
( time python -c "import torch" ) > gpurun_out/rehearse_import.log 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --device 0 \
  --steps 20 --warmup 5 > gpurun_out/rehearsal_gloo_2ranks_one_gpu.json 2> gpurun_out/rehearse_err.log || { tail -20 gpurun_out/rehearse_err.log; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/rehearsal_gloo_2ranks_one_gpu.json"))
print("n_gpus", d["n_gpus"], "value %.4g" % d["value"], "parity_check", d["ranks"]["parity_check"]["ok"], d["ranks"]["parity_check"]["reads"], "efficiency", round(d["efficiency_vs_n1_reference"], 3))
PY
