#!/bin/bash
# A/B of benchmark variants selected by environment variables, interleaved in rounds.
# usage: tools/ab_env.sh <tag> "<bench args>" "VAR=val ..." "VAR=val ..." ...
set -o pipefail
tag=$1; bargs=$2; shift 2
out=gpurun_out/ab_$tag.txt
mkdir -p gpurun_out; : > $out
for round in 1 2 3; do
  i=0
  for v in "$@"; do
    echo "## variant $i [$v] round $round" >> $out
    env $v timeout -k 10 120 python3 bench.py $bargs --no-cpu-baseline --no-other-configs >> $out 2>&1 || { echo "variant failed: $v"; tail -5 $out; exit 1; }
    i=$((i+1))
  done
done
python3 - <<PY
import json, collections
cur=None; res=collections.defaultdict(list); ker=collections.defaultdict(list)
for l in open("$out"):
    if l.startswith("## variant"): cur=l.split("[")[1].split("]")[0]
    elif l.startswith("{"):
        d=json.loads(l); res[cur].append(d["value"])
        if d.get("roofline"): ker[cur].append({k: round(v["avg_ms"]*1e3) for k,v in d["roofline"]["kernels"].items()})
for k,v in res.items():
    print(f"{k:50s} median {sorted(v)[len(v)//2]:10.0f}  all {[round(x) for x in v]}  kernels(us) {ker[k][-1] if ker[k] else None}")
PY
