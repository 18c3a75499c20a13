#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into small tracked files under profiles/.
usage: summarize_prof.py <rocprof_out_dir> <profiles/out_prefix>"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", name)
    if m:
        return m.group(1)
    return name[:60]


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    for f in glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        with open(prefix + "_kernel_stats.csv", "w") as o:
            o.write("kernel,calls,avg_ns,min_ns,max_ns,percent\n")
            for r in rows:
                o.write("%s,%s,%.1f,%s,%s,%s\n" % (short(r["Name"]).replace(",", ";"), r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
    for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            key = "%s grid=%sx%s" % (short(r["Kernel_Name"]), r["Grid_Size_X"], r["Grid_Size_Y"])
            by[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        out = {k: {"launches": len(v), "avg_ns": sum(v) / len(v), "min_ns": min(v), "max_ns": max(v)} for k, v in by.items() if k.startswith("k_")}
        json.dump(out, open(prefix + "_kernel_trace_by_grid.json", "w"), indent=1, sort_keys=True)
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = short(r["Kernel_Name"])
            if "Grid_Size" in r and name.startswith("k_nn_red"):
                name += " grid=%s" % r["Grid_Size"]        # separates the 32-scans-in-flight launches from single-scan ones
            agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        out = {}
        for (k, c), v in sorted(agg.items()):
            out.setdefault(k, {})[c] = {"dispatches": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v)}
        json.dump(out, open(prefix + "_pmc.json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
