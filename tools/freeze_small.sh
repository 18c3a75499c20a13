#!/bin/bash
# frozen pairs below the automatic threshold: scans/s for small batches, automatic rule vs forced (bench.py --force-freeze)
cd "$GRAFT_REPO_ROOT"
for b in 1 2 4 8; do
  for f in "" "--force-freeze"; do
    echo -n "batch $b $f: "
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --batch $b --steps 40 $f 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],3))'
  done
done
