#!/bin/bash
# The bench lines and stream-demo lines kept under profiles/ (one gpurun call): writes gpurun_out/profiles_<P>/<P>_bench_*.json,
# <P>_bench_sweep.jsonl and <P>_stream_demo.jsonl.   tools/collect_bench.sh r02
set -e
P=${1:-r03}
cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/profiles_$P"
mkdir -p "$OUT"
line() { grep '"metric"' | tail -1; }
timeout -k 10 300 python3 bench.py 2>/dev/null | line > "$OUT/${P}_bench_default.json"; echo default
timeout -k 10 300 python3 bench.py --mode ref_cpp 2>/dev/null | line > "$OUT/${P}_bench_ref_cpp.json"; echo ref_cpp
timeout -k 10 300 python3 bench.py --mode o3d_p2p 2>/dev/null | line > "$OUT/${P}_bench_o3d_p2p.json"; echo o3d
timeout -k 10 300 python3 bench.py --no-nn-reuse --no-cpu-baseline --no-extras 2>/dev/null | line > "$OUT/${P}_bench_search.json"; echo search
timeout -k 10 300 python3 bench.py --force-dist --no-cpu-baseline --no-extras 2>/dev/null | line > "$OUT/${P}_bench_forcedist.json"; echo forcedist
timeout -k 10 300 python3 bench.py --force-dist --collective torch --no-cpu-baseline --no-extras 2>/dev/null | line > "$OUT/${P}_bench_forcedist_torch.json"; echo forcedist_torch
: > "$OUT/${P}_bench_sweep.jsonl"
for b in 1 2 4 8 16 32 128; do timeout -k 10 300 python3 bench.py --batch $b --no-cpu-baseline --no-extras 2>/dev/null | line >> "$OUT/${P}_bench_sweep.jsonl"; done; echo sweep
: > "$OUT/${P}_stream_demo.jsonl"
timeout -k 10 300 python3 tools/stream_demo.py --scans 1000 2>/dev/null | tail -1 >> "$OUT/${P}_stream_demo.jsonl"
timeout -k 10 300 python3 tools/stream_demo.py --scans 1000 --python-flow 2>/dev/null | tail -1 >> "$OUT/${P}_stream_demo.jsonl"
timeout -k 10 300 python3 tools/stream_demo.py --scans 1000 --prior ekf 2>/dev/null | tail -1 >> "$OUT/${P}_stream_demo.jsonl"
timeout -k 10 300 python3 tools/stream_demo.py --scans 1000 --prior imu-ekf-growth 2>/dev/null | tail -1 >> "$OUT/${P}_stream_demo.jsonl"
echo stream
