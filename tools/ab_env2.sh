#!/bin/bash
# A/B of environment switches of the library, several at once:  tools/ab_env2.sh "SF_UPLOAD_STAGGER=0" "SF_UPLOAD_STAGGER=1" ...
# per setting: the bench line's value (same source every step, every result fetched) and the streaming probe (a new batch from pinned host memory every step)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for e in "$@"; do
  echo -n "[$e] bench: "
  env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],3), d["parity"]["ok"], end="  ")'
  echo -n " stream probe: "
  env $e timeout -k 10 300 python3 tools/probes/stream_timing.py 12 2>/dev/null | grep -v "  set_source" | tr '\n' ' '
  echo
done
done
