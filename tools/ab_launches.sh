#!/bin/bash
# A/B of library variants in one gpurun call, per-launch view:  tools/ab_launches.sh lib1.so lib2.so ...
# prints scans/s and the k_nn_red launch durations of one alignment (bench.py --no-extras, parity gate ignored: stub builds fail it)
L=slam_sensor_fusion_amd/lib/libslamfusion.so
cp $L /tmp/orig.so
for rep in 1 2; do
for lib in "$@"; do
  cp $lib $L
  python bench.py $BENCH_EXTRA --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], 'scans/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'parity', d['parity']['ok'])
print('   us', r['per_launch_us'])
print('   search frac', r['per_launch_queries_searching_frac'])" $(basename $lib)
done
done
cp /tmp/orig.so $L
