#!/bin/bash
# one rocprofv3 --pmc pass over the benchmark: tools/pmc_pass.sh <outdir-name> "<counters>" [bench args]
set -o pipefail
export TMPDIR=/tmp
name=$1; counters=$2; shift 2
out=$PWD/gpurun_out/$name
mkdir -p "$out"
timeout -s KILL 200 rocprofv3 --pmc $counters --output-format csv -d "$out" -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-events --no-other-configs "$@" > "$out/run.log" 2>&1 || { echo "pmc pass failed"; tail -5 "$out/run.log"; exit 1; }
tail -2 "$out/run.log"
