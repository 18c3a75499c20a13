#!/usr/bin/env python3
"""Timeline of the last kernels / copies of a rocprofv3 --kernel-trace --memory-copy-trace run: tools/probes/timeline.py DIR [window_ms]"""
import csv, glob, sys
d = sys.argv[1]; win = float(sys.argv[2]) if len(sys.argv) > 2 else 14.0; skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:30], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:28], "copy"))
ev.sort()
t_end = max(e[1] for e in ev)
big = [e for e in ev if t_end - (win + skip) * 1e6 < e[1] <= t_end - skip * 1e6 and (e[1] - e[0] > 60e3 or e[2].startswith("COPY"))]
t0 = big[0][0]
for s, e, n, q in big:
    print("%9.3f %9.3f  %7.3f ms  %-5s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
