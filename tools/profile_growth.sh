#!/bin/bash
# Kernel trace of the growth step (tools/growth_bench.py, 4 timed steps per variant) -> gpurun_out/profiles_<P>/<P>_growth_kernel_stats.csv
set -e
P=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/profiles_$P"
W=/tmp/prof_growth_$P
mkdir -p "$OUT" "$W"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$W" -o t --output-format csv -- python3 tools/growth_bench.py --steps 4 > "$OUT/growth.log" 2>&1
python3 tools/summarize_prof.py "$W" "$OUT/${P}_growth"
