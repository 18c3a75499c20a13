#!/bin/bash
# vector instructions per wave and duration of the dominant kernel for the library as built (one PMC pass + one timing pass):
#   tools/ab_valu.sh <label> [bench.py arguments]
set -e
L=${1:-x}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=/tmp/abv_$L; mkdir -p $W gpurun_out/abv
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-extras --no-pipeline $*"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $W/sq -o t --output-format csv -- $CMD > /dev/null 2>&1
python3 tools/summarize_prof.py $W/sq gpurun_out/abv/${L}_sq > /dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $W/tm -o t --output-format csv -- $CMD > /dev/null 2>&1
python3 tools/summarize_prof.py $W/tm gpurun_out/abv/${L}_tm > /dev/null
python3 - <<PY
import json, csv
d = json.load(open("gpurun_out/abv/${L}_sq_pmc.json"))
k = max((k for k in d if k.startswith("k_nn_red") or k.startswith("k_ref_nn")), key=lambda k: d[k]["SQ_WAVES"]["avg"])
w = d[k]["SQ_WAVES"]["avg"]
print("${L}", k, "valu/wave %.0f salu %.0f lds %.0f vmem %.0f" % tuple(d[k][c]["avg"] / w for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD")))
for r in csv.DictReader(open("gpurun_out/abv/${L}_tm_kernel_stats.csv")):
    if r["kernel"].startswith("k_nn_red") or r["kernel"].startswith("k_ref_nn"):
        print("   ", r["kernel"], "avg us %.1f min %.1f max %.1f" % (float(r["avg_ns"]) / 1e3, float(r["min_ns"]) / 1e3, float(r["max_ns"]) / 1e3))
PY
