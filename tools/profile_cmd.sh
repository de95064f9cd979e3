#!/bin/bash
# Runs on the GPU box: kernel trace and counter passes of any python workload script of this repository.
# usage: tools/profile_cmd.sh <tag> <script.py> [args...]      outputs under gpurun_out/prof_<tag>/
# One rocprofv3 run per counter block (the hardware cannot collect them together); the program follows "--" directly.
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$@" > "$out/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$out/trace.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$out/pmc_sq" -- python3 "$@" > "$out/pmc_sq.log" 2>&1 || { echo "pmc sq failed"; tail -5 "$out/pmc_sq.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 "$@" > "$out/pmc_fetch.log" 2>&1 || { echo "pmc fetch failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 "$@" > "$out/pmc_write.log" 2>&1 || { echo "pmc write failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d "$out/pmc_sq2" -- python3 "$@" > "$out/pmc_sq2.log" 2>&1 || { echo "pmc sq2 failed"; }
tail -2 "$out/trace.log"
echo done
