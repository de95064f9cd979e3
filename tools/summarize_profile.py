#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh output directory into the files kept under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (library kernels only)
  <tag>_pmc.json           per-kernel averages of every collected counter
  traffic_<tag>.json       HBM bytes per launch per kernel, from FETCH_SIZE / WRITE_SIZE with the
                           gfx950 corrections: both counters are in KiB; FETCH_SIZE tallies 128-byte
                           requests of wide coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE
                           is exact.
usage: tools/summarize_profile.py gpurun_out/prof_<tag> <tag> [batch per launch, default 256] [n, default 131072]"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"nttamd::(k_\w+)", name)
    return m.group(1) if m else None


def main(src, tag, batch=256, n=131072):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(root, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(rows[0])
            for r in rows[1:]:
                if short(r[0]):
                    w.writerow(r)
    # per-launch durations over time since the first launch: the clock transient and the steady state after it
    traces = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
    if traces:
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(traces[0])):
            k = short(r["Kernel_Name"])
            if k:
                per[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        t0 = min(v[0][0] for v in per.values() if v) if per else 0
        steady = {}
        for k, v in per.items():
            v.sort()
            early = [d for s_, d in v if s_ - t0 < 30e6]
            late = [d for s_, d in v if s_ - t0 >= 60e6]
            steady[k] = {"launches": len(v),
                         "avg_us_first_30ms": sum(early) / len(early) / 1e3 if early else None,
                         "max_us_first_30ms": max(early) / 1e3 if early else None,
                         "avg_us_after_60ms": sum(late) / len(late) / 1e3 if late else None,
                         "launches_after_60ms": len(late)}
        json.dump(steady, open(os.path.join(dst, f"{tag}_clock_transient.json"), "w"), indent=1, sort_keys=True)
        # the same summary as rocprofv3's kernel_stats, over the launches that start 60 ms or more after the first one only
        # (the steady state: the clock transient of the first ~25 ms is left out)
        with open(os.path.join(dst, f"{tag}_kernel_stats_steady.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Note"])
            for k, v in sorted(per.items()):
                late = [d for s_, d in v if s_ - t0 >= 60e6]
                if late:
                    w.writerow([k, len(late), sum(late), sum(late) / len(late), min(late), max(late),
                                "launches starting >= 60 ms after the first launch of the trace"])
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k:
                pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                pmc[k]["VGPR_Count"] = [float(r["VGPR_Count"])]
                pmc[k]["LDS_Block_Size"] = [float(r["LDS_Block_Size"])]
    summary = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
    json.dump(summary, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
    traffic = {}
    for k, d in summary.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            rd, wr = d["FETCH_SIZE"] * 1024 * 2, d["WRITE_SIZE"] * 1024
            traffic[k] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
    traffic["_meta"] = {"batch": batch, "n": n, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes"}
    json.dump(traffic, open(os.path.join(dst, f"traffic_{tag}.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 256, int(sys.argv[4]) if len(sys.argv) > 4 else 131072)
