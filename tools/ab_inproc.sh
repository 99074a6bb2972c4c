#!/bin/bash
# tools/r4_inproc.sh <tag> <ab_inproc args...>   (several runs separated by ';;')
set -o pipefail
mkdir -p gpurun_out
TAG=$1; shift
( time python -c "import torch" ) > gpurun_out/${TAG}_import.log 2>&1
rm -f gpurun_out/${TAG}.log
args=()
for a in "$@" ";;"; do
  if [ "$a" = ";;" ]; then
    echo "== ${args[*]}" >> gpurun_out/${TAG}.log
    timeout -k 10 600 python tools/ab_inproc.py "${args[@]}" >> gpurun_out/${TAG}.log 2>gpurun_out/${TAG}_err.log || { tail -20 gpurun_out/${TAG}_err.log; exit 1; }
    args=()
  else
    args+=("$a")
  fi
done
cat gpurun_out/${TAG}.log
