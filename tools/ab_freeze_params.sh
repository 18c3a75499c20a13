#!/bin/bash
# A/B of freeze parameter sets on the bench line (same build):  tools/ab_freeze_params.sh "" "8,2e-5,1e-3,3,4" ...
# prints value, ms per step, the first launches' durations and the freeze statistics for every set, twice
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for p in "$@"; do
  echo -n "[$p] "
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras ${p:+--freeze-params "$p"} $BENCH_EXTRA 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
r=d.get("roofline") or {}
fz=r.get("frozen_pairs") or {}
print(round(d["value"]), round(d["ms_per_step"],3), "parity", d["parity"]["ok"], [round(x) for x in (r.get("per_launch_us") or [])[:9]], {k: fz.get(k) for k in ("froze","frozen_at_end","active_queries","failed","thawed","freeze_launch_us","frozen_launch_us")})'
done
done
