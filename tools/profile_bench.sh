#!/bin/bash
# Runs on the GPU box: kernel-trace stats and PMC passes of the benchmark.  Outputs under gpurun_out/prof_<tag>/.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
tag=${1:-r01}; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
# the trace run keeps the benchmark's clock ramp (so its per-kernel averages are steady-state ones, like the
# benchmark's own); the counter passes do not need it
args="--steps 5 --warmup 2 --ramp-ms 0 --no-cpu-baseline --no-kernel-events --no-other-configs $*"
targs="--steps 100 --warmup 10 --no-cpu-baseline --no-kernel-events --no-other-configs $*"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py $targs > "$out/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$out/trace.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$out/pmc_sq" -- python3 bench.py $args > "$out/pmc_sq.log" 2>&1 || { echo "pmc sq failed"; tail -5 "$out/pmc_sq.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py $args > "$out/pmc_fetch.log" 2>&1 || { echo "pmc fetch failed"; tail -5 "$out/pmc_fetch.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py $args > "$out/pmc_write.log" 2>&1 || { echo "pmc write failed"; tail -5 "$out/pmc_write.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d "$out/pmc_sq2" -- python3 bench.py $args > "$out/pmc_sq2.log" 2>&1 || { echo "pmc sq2 failed"; tail -5 "$out/pmc_sq2.log"; }
find "$out" -name "*.csv" | head -30
echo done
