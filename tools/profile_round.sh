#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box and write the summaries under gpurun_out/profiles_<P>/
# (copy what is to be judged into profiles/):
#   tools/profile_round.sh r02                          the default bench workload (p2plane, 64 scans in flight, reuse on)
#   tools/profile_round.sh r02_search --no-nn-reuse     extra bench.py arguments: here every query searches in every launch
#   tools/profile_round.sh r02_refcpp --mode ref_cpp
# Timing pass: --kernel-trace --stats.  Counter passes: --pmc only, one group per run (gpurun refuses --pmc together with
# tracing flags).  Command profiled: bench.py, 3 steps, no graph replay (kernels visible one by one), no extra legs.
set -e
P=${1:-r03}
shift || true
EXTRA="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/profiles_$P"
W=/tmp/prof_$P
mkdir -p "$OUT" "$W"
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-extras --no-pipeline $EXTRA"   # (one lane: kernels traced one at a time, not two alignments side by side)
run() { # name, rocprofv3 args...
    local name=$1; shift
    timeout -k 10 200 rocprofv3 "$@" -d "$W/$name" -o t --output-format csv -- $CMD > "$OUT/$name.log" 2>&1
    python3 tools/summarize_prof.py "$W/$name" "$OUT/${P}_$name"
    echo "$name done"
}
run final --kernel-trace --stats
if [ "$PASSES" != "timing" ]; then
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run rdsz --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run wrsz --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum
run mem --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
fi
if [ "$PASSES" == "all" ]; then
# texture-path units (at most two TA / TD counters fit one pass)
run ta --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run td --pmc TD_TD_BUSY_sum TD_TC_STALL_sum
run tcp --pmc TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
fi
rm -f "$OUT"/*.log
ls "$OUT"
