#!/bin/bash
# A/B of library variants in one gpurun call: tools/ab_bench.sh "B1 B2 .." lib1.so lib2.so ...
L=slam_sensor_fusion_amd/lib/libslamfusion.so
BS="$1"; shift
cp $L /tmp/orig.so
for rep in 1 2; do
for lib in "$@"; do
  cp $lib $L
  for B in $BS; do
    python bench.py $BENCH_EXTRA --steps 8 --warmup 2 --batch $B --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], 'B', d['config']['scans_in_flight'], 'scans/s', round(d['value'],1), 'nn_us', round(d['roofline']['avg_launch_ms']*1e3,1), d['parity']['ok'])" $(basename $lib)
  done
done
done
cp /tmp/orig.so $L
