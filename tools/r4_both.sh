#!/bin/bash
# tests (filter $1), then ab_inproc runs given as further args separated by ';;'
set -o pipefail
mkdir -p gpurun_out
( time python -c "import torch" ) > gpurun_out/r4_both_import.log 2>&1
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -v -x -k "$1" > gpurun_out/r4_both_tests.log 2>&1 || { tail -40 gpurun_out/r4_both_tests.log; exit 1; }
tail -2 gpurun_out/r4_both_tests.log
shift
bash tools/r4_inproc.sh r4_both_ab "$@"
