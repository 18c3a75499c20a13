cd "$GRAFT_REPO_ROOT"
for c in 0.2 0.25 0.3 0.35 0.4; do
  echo -n "cell $c: "
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --cell $c 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d["roofline"]
print(round(d["value"]), round(d["ms_per_step"],3), [round(x) for x in (r.get("per_launch_us") or [])[:6]])'
done
