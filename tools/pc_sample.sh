#!/bin/bash
# PC sampling of the library as built (stochastic, hardware):  tools/pc_sample.sh <label> [bench.py arguments]
# -> gpurun_out/pcs/<label>_top.txt (hottest instructions / source lines of the dominant kernel)
L=${1:-x}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=/tmp/pcs_$L; rm -rf $W; mkdir -p $W gpurun_out/pcs
METHOD=${PCS_METHOD:-stochastic}; UNIT=${PCS_UNIT:-cycles}; IVL=${PCS_INTERVAL:-65536}
timeout -k 10 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $IVL --kernel-trace \
  -d $W -o t --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-extras "$@" > gpurun_out/pcs/${L}_run.log 2>&1
echo "rocprofv3 exit $?" >> gpurun_out/pcs/${L}_run.log
find $W -type f | head -20 >> gpurun_out/pcs/${L}_run.log
python3 tools/pc_sample_summary.py $W gpurun_out/pcs/${L} >> gpurun_out/pcs/${L}_run.log 2>&1
