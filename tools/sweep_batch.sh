#!/bin/bash
# batch-size sweep of the benchmark (one JSON line per batch size) -> gpurun_out/sweep_<tag>.txt
set -o pipefail
tag=${1:-r01}
out=gpurun_out/sweep_$tag.txt
: > $out
for b in 8 16 32 64 128 256 512; do
  timeout -k 10 120 python3 bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs >> $out 2>&1 || exit 1
done
python3 - <<PY
import json
for l in open("$out"):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]["kernels"]
        print(d["config"]["batch_per_gpu"], round(d["value"]), {k: round(v["avg_ms"]*1e3/d["config"]["batch_per_gpu"],3) for k,v in r.items()})
PY
