mkdir -p gpurun_out; : > gpurun_out/fuzz.log
( time python -c "import torch" ) > gpurun_out/fuzz_import.log 2>&1
timeout -k 10 400 python tools/fuzz_gpu.py --seconds 200 --seed 4001 >> gpurun_out/fuzz.log 2>&1 || { tail -5 gpurun_out/fuzz.log; exit 1; }
for e in QUACK_HIP_TUNE=group=3 QUACK_HIP_TUNE=small_ring QUACK_HIP_NO_NEUTRAL=1 QUACK_HIP_TUNE=pad_always; do
  echo "== $e" >> gpurun_out/fuzz.log
  env $e timeout -k 10 200 python tools/fuzz_gpu.py --seconds 60 --seed 9001 >> gpurun_out/fuzz.log 2>&1 || { tail -5 gpurun_out/fuzz.log; exit 1; }
done
grep -E "fuzz:|==|MISMATCH" gpurun_out/fuzz.log
