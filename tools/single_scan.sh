#!/bin/bash
# one 200 k-point scan at a time (the reference's calling pattern): per-launch times, ms per alignment, and the bench line's figures
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --batch 1 --no-pipeline --steps 50 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d["roofline"]
print("batch 1, one lane:", round(d["value"]), "scans/s", round(d["ms_per_step"],4), "ms per alignment; launches us", [round(x,1) for x in (r.get("per_launch_us") or [])])'
timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print({k: (round(d[k],4) if isinstance(d[k],float) else d[k]) for k in ("value","ms_per_step","value_no_pipeline","single_scan_latency_ms","value_no_freeze","value_no_reuse")}, d["parity"])'
