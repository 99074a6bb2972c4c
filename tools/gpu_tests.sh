#!/bin/bash
# the whole -m gpu suite, verbose into gpurun_out/ (progress for the silence watchdog)
set -o pipefail
mkdir -p gpurun_out
( time python -c "import torch" ) > gpurun_out/r4_full_import.log 2>&1
if [ $# -eq 0 ]; then set -- tests; fi
timeout -k 10 1100 python -m pytest "$@" -v -x -m gpu > gpurun_out/r4_full_tests.log 2>&1; rc=$?
grep -E "passed|failed|error" gpurun_out/r4_full_tests.log | tail -3
[ $rc -eq 0 ] || tail -40 gpurun_out/r4_full_tests.log
exit $rc
