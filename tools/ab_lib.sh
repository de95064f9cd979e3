#!/bin/bash
# A/B of library builds (diagnostic variants vs shipped), interleaved rounds, per-kernel event times.
# usage: tools/ab_lib.sh <tag> "<bench args>" lib1.so lib2.so ...   ("shipped" = the package library)
set -o pipefail
tag=$1; bargs=$2; shift 2
out=gpurun_out/ablib_$tag.txt
mkdir -p gpurun_out; : > $out
for round in 1 2 3; do
  for v in "$@"; do
    echo "## variant [$v] round $round" >> $out
    if [ "$v" = shipped ]; then la=""; else la="--lib $v --no-verify"; fi
    timeout -k 10 120 python3 bench.py $bargs $la --no-cpu-baseline --no-other-configs >> $out 2>&1 || { echo "variant failed: $v"; tail -5 $out; exit 1; }
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
    print(f"{k:44s} median {sorted(v)[len(v)//2]:10.0f}  kernels(us) {ker[k]}")
PY
