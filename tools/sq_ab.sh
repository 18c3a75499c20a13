#!/bin/bash
# Vector / scalar / memory instruction counts per wave of k_nn_red for each library variant given (rocprofv3 --pmc, one pass
# each):  BENCH_EXTRA=--no-nn-reuse tools/sq_ab.sh libA.so libB.so   (the x-sorted-cells row)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
L=slam_sensor_fusion_amd/lib/libslamfusion.so
cp $L /tmp/orig.so
for lib in "$@"; do
  cp $lib $L
  rm -rf /tmp/sq_$$
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d /tmp/sq_$$ -o t --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-extras --no-pipeline $BENCH_EXTRA > /dev/null 2>&1
  python3 tools/summarize_prof.py /tmp/sq_$$ /tmp/sq_$$/sum
  python3 - <<PY
import json
d=json.load(open("/tmp/sq_$$/sum_pmc.json"))
for k,v in d.items():
    if k.startswith("k_nn_red") and v["SQ_WAVES"]["avg"]>50000:
        w=v["SQ_WAVES"]["avg"]
        print("$(basename $lib)", k[:40], "valu/wave %.0f salu %.0f vmem_rd %.1f lds %.1f" % (v["SQ_INSTS_VALU"]["avg"]/w, v["SQ_INSTS_SALU"]["avg"]/w, v["SQ_INSTS_VMEM_RD"]["avg"]/w, v["SQ_INSTS_LDS"]["avg"]/w))
PY
done
cp /tmp/orig.so $L
