#!/usr/bin/env python3
"""Device timeline of one per-scan callback from a rocprofv3 trace of tools/stream_demo.py
(rocprofv3 --kernel-trace --memory-copy-trace -d DIR -o t --output-format csv -- python3 tools/stream_demo.py --scans 120):
   tools/stream_timeline.py DIR [callbacks to print]"""
import csv
import sys

d = sys.argv[1]
n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for r in csv.DictReader(open(d + "/t_kernel_trace.csv")):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:48]))
try:
    for r in csv.DictReader(open(d + "/t_memory_copy_trace.csv")):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "").replace("MEMORY_COPY_", "")))
except FileNotFoundError:
    pass
rows.sort()
marks = [i for i, r in enumerate(rows) if "k_prep_count" in r[2] or "k_subsample" in r[2]]
for s in marks[-(n_show + 1):-1]:
    i0 = s
    while i0 > 0 and rows[i0 - 1][2].startswith("COPY") and rows[s][0] - rows[i0 - 1][0] < 500000:
        i0 -= 1
    t0 = rows[i0][0]
    print("---- one callback (us from the start of the scan's upload: start, duration, what)")
    j = i0
    while j < len(rows) and rows[j][0] - t0 < 1500000 and (j <= s or not ("k_prep_count" in rows[j][2] or "k_subsample" in rows[j][2])):
        a, b, n = rows[j]
        print("%9.2f  %8.2f  %s" % ((a - t0) / 1000.0, (b - a) / 1000.0, n))
        j += 1
