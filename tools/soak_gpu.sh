#!/bin/bash
# Runs on the GPU box: the seeded random sweeps of tests/test_gpu_fuzz.py with other seeds and more cases, one pytest
# process per seed, one after the other; stops at the first failure.
# usage: tools/soak_gpu.sh <scale> <seed> [<seed> ...]      log under gpurun_out/soak.log
set -o pipefail
scale=$1; shift
mkdir -p gpurun_out
: > gpurun_out/soak.log
for seed in "$@"; do
  echo "== seed $seed scale $scale" >> gpurun_out/soak.log
  NTT_FUZZ_SEED=$seed NTT_FUZZ_SCALE=$scale timeout -k 10 900 python3 -u -m pytest tests/test_gpu_fuzz.py -x -q --timeout 800 --timeout-method thread >> gpurun_out/soak.log 2>&1 || { echo "seed $seed failed"; tail -30 gpurun_out/soak.log; exit 1; }
  tail -1 gpurun_out/soak.log
done
echo soak done
