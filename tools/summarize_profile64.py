#!/usr/bin/env python3
"""Condenses a tools/profile_config5.sh output directory into profiles/<tag>_kernel_stats.csv (library kernels of the
rocprofv3 --kernel-trace --stats summary) and profiles/<tag>_pmc.json (per-kernel averages of every counter, HBM bytes per
launch from FETCH_SIZE / WRITE_SIZE with the gfx950 corrections: KiB units, FETCH_SIZE doubled).  Kernels are keyed by
name and template arguments, e.g. k64_tile<FieldD,fwd,raw>.
usage: tools/summarize_profile64.py gpurun_out/prof_<tag> <tag>"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"nttamd::(k64_\w+)(<[^>]*>)?", name)
    if not m:
        return None
    args = (m.group(2) or "").replace("nttamd::", "").replace(" ", "")
    return m.group(1) + args


def main(src, tag):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(root, "profiles")
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(rows[0])
            for r in rows[1:]:
                if short(r[0]):
                    w.writerow([short(r[0])] + r[1:])
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k:
                pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                pmc[k]["VGPR_Count"] = [float(r["VGPR_Count"])]
                pmc[k]["LDS_Block_Size"] = [float(r["LDS_Block_Size"])]
    summary = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
    for k, d in summary.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_read_bytes"] = d["FETCH_SIZE"] * 1024 * 2
            d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    json.dump(summary, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
    times = {}
    if stats:
        for r in csv.DictReader(open(stats[0])):
            if short(r["Name"]):
                times[short(r["Name"])] = float(r["AverageNs"]) / 1e3
    for k, d in sorted(summary.items()):
        t = times.get(k)
        line = f"{k:45s} {t or 0:8.1f} us"
        if "hbm_read_bytes" in d:
            line += f"  rd {d['hbm_read_bytes']/2**30:5.2f} GiB wr {d['hbm_write_bytes']/2**30:5.2f} GiB"
            if t:
                line += f"  {(d['hbm_read_bytes'] + d['hbm_write_bytes']) / (t * 1e-6) / 1e12:4.2f} TB/s"
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"):
            if c in d:
                line += f"  {c[3:]}={d[c]:.3g}"
        print(line)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
