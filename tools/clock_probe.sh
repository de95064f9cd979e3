#!/bin/bash
# Average shader clock per kernel = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, for one or
# more builds of the library.  usage: tools/clock_probe.sh [lib.so ...]   (no argument: the shipped build)
set -o pipefail
export TMPDIR=/tmp
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(shipped)
for lib in "${libs[@]}"; do
  name=$(basename "$lib" .so)
  out=$PWD/gpurun_out/clock_$name
  mkdir -p "$out"
  if [ "$lib" = shipped ]; then la=""; else la="--lib $lib --no-verify"; fi
  timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out" -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-events --no-other-configs $la > "$out/run.log" 2>&1 || { echo "failed: $lib"; tail -5 "$out/run.log"; exit 1; }
  python3 - "$out" "$name" <<'PY'
import csv, glob, sys, collections, re
out, name = sys.argv[1], sys.argv[2]
cc = glob.glob(out + "/*/*_counter_collection.csv")[0]
kt = glob.glob(out + "/*/*_kernel_trace.csv")[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    m = re.search(r"nttamd::(k_\w+)", r["Kernel_Name"])
    if m and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
        acc[m.group(1)].append((float(r["Counter_Value"]) / 8 / dur[r["Dispatch_Id"]], dur[r["Dispatch_Id"]] / 1e3))
for k, v in acc.items():
    tail = v[len(v) // 2:]   # second half of the run: past the clock transient
    print(name, k, "GHz %.3f" % (sum(x[0] for x in tail) / len(tail)), "us %.1f" % (sum(x[1] for x in tail) / len(tail)), "launches", len(v))
PY
done
