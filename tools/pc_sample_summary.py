#!/usr/bin/env python3
"""Summarise a rocprofv3 PC-sampling run: samples per kernel, per instruction and per source line.
usage: pc_sample_summary.py <rocprof output dir> <output prefix>"""
import collections
import csv
import glob
import os
import sys

csv.field_size_limit(1 << 30)
src, out = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(src, "**", "*pc_sampling*.csv"), recursive=True)
kern = {}
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kern[r.get("Dispatch_Id")] = r.get("Kernel_Name", "?")
print("pc sampling files:", files)
per_kernel = collections.Counter()
per_inst = collections.defaultdict(collections.Counter)
per_line = collections.defaultdict(collections.Counter)
extra_cols = None
n = 0
for f in files:
    with open(f) as fh:
        rd = csv.DictReader(fh)
        if extra_cols is None:
            extra_cols = rd.fieldnames
            print("columns:", extra_cols)
        for r in rd:
            n += 1
            k = kern.get(r.get("Dispatch_Id"), "?")
            k = k.split("(")[0][-60:]
            per_kernel[k] += 1
            per_inst[k][r.get("Instruction", "?")] += 1
            per_line[k][r.get("Instruction_Comment", "?")] += 1
print("samples:", n)
with open(out + "_top.txt", "w") as o:
    for k, c in per_kernel.most_common(6):
        o.write("=== %s: %d samples (%.1f %%)\n" % (k, c, 100.0 * c / max(n, 1)))
        o.write("--- by source line\n")
        for line, cc in per_line[k].most_common(60):
            o.write("%7d %5.1f%%  %s\n" % (cc, 100.0 * cc / c, line))
        o.write("--- by instruction text\n")
        for ins, cc in per_inst[k].most_common(80):
            o.write("%7d %5.1f%%  %s\n" % (cc, 100.0 * cc / c, ins))
print(open(out + "_top.txt").read()[:3000])
