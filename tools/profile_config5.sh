#!/bin/bash
# Runs on the GPU box: kernel trace and counter passes of the single n = 2^28 product (tools/bench_config5.py).
# usage: tools/profile_config5.sh <tag> [bench_config5 args...]      outputs under gpurun_out/prof_<tag>/
set -o pipefail
tag=${1:-c5}; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
args="--direct-only --reps 2 $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 tools/bench_config5.py $args > "$out/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$out/trace.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$out/pmc_sq" -- python3 tools/bench_config5.py $args > "$out/pmc_sq.log" 2>&1 || { echo "pmc sq failed"; tail -5 "$out/pmc_sq.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 tools/bench_config5.py $args > "$out/pmc_fetch.log" 2>&1 || { echo "pmc fetch failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 tools/bench_config5.py $args > "$out/pmc_write.log" 2>&1 || { echo "pmc write failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d "$out/pmc_sq2" -- python3 tools/bench_config5.py $args > "$out/pmc_sq2.log" 2>&1 || { echo "pmc sq2 failed"; }
find "$out" -name "*.csv" | head -30
echo done
