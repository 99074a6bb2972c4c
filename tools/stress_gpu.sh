#!/bin/bash
# Run ON THE GPU BOX: the parity suite again under launch geometries the planner
# would not pick by itself (small tiles -> everything multi-tile, other
# workgroup sizes, forced pipelining, separate adapter kernels, 1 MiB slots).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out; : > gpurun_out/stress.log
( time python -c 'import torch' ) > gpurun_out/stress_import.log 2>&1
# (strided batches run under every override too: the shim restates them as gapped batches when the strided
# kernel variant is not built for the geometry)
K="not cli and not launch_configurations and not pipelined and not tuning_overrides and not under_overrides"
for e in "QUACK_HIP_TILE=64" "QUACK_HIP_THREADS=512" "QUACK_HIP_TILE=128 QUACK_HIP_THREADS=256 QUACK_HIP_UNROLL=2" \
         "QUACK_HIP_PIPE=2 QUACK_HIP_UNROLL=2" "QUACK_HIP_UNFUSED_ADAPTERS=1" "QUACK_HIP_REPLICAS=1" "QUACK_HIP_REPLICAS=2" \
         "QUACK_HIP_ADAPT_PD=3" "QUACK_HIP_ADAPT_PD=4 QUACK_HIP_ADAPT_U=1" "QUACK_HIP_SEPARATE_COUNT=1" \
         "QUACK_HIP_NO_GROUP=1" "QUACK_HIP_GROUP=2" "QUACK_HIP_SMALL_RING=1" "QUACK_HIP_RING_WORDS=4096" "QUACK_HIP_NO_PAD=1" "QUACK_HIP_PAD_ALWAYS=1" \
         "QUACK_HIP_NO_W16=1" "QUACK_HIP_NO_SIDE=1" "QUACK_HIP_NO_TABLE32=1" "QUACK_HIP_TILE_OVERHEAD=0" "QUACK_HIP_LENGTH_KERNEL=1"; do
  echo "== $e" | tee -a gpurun_out/stress.log
  env $e timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -rf -k "$K" 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-220 | tee -a gpurun_out/stress.log
done
echo "== QUACK_HIP_BATCH_MB=1"
QUACK_HIP_BATCH_MB=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -rf -k "$K and not gapped and not promise" 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-220
