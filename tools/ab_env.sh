#!/bin/bash
# A/B of an environment switch of the library on the bench line (same build): tools/ab_env.sh VAR [bench args...]
# prints value, ms per step, per-launch times for VAR=0 and VAR=1
cd "$GRAFT_REPO_ROOT"
V=$1; shift
for t in 0 1 0 1; do
  echo -n "$V=$t: "
  env $V=$t timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
r=d.get("roofline") or {}
print(round(d["value"]), round(d["ms_per_step"],3), [round(x) for x in (r.get("per_launch_us") or [])[:8]])'
done
