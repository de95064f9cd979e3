#!/bin/bash
# Runs on the GPU box: kernel trace of a long benchmark run (per-launch durations over time) with
# clock/power samples taken beside it.  Outputs under gpurun_out/throttle/.
set -o pipefail
export TMPDIR=/tmp
out=$PWD/gpurun_out/throttle
mkdir -p "$out"
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "sclk|Power|Temperature \(Sensor (edge|junction)" ; echo "--- $i"; sleep 0.5; done ) > "$out/smi.log" 2>&1 &
smi=$!
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 bench.py --steps 3000 --warmup 2 --no-cpu-baseline --no-kernel-events > "$out/trace.log" 2>&1
rc=$?
kill $smi 2>/dev/null
tail -2 "$out/trace.log"
exit $rc
