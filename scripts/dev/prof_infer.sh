#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (3 in flight) and of --inflight 1 (clean per-kernel times).
# usage (GPU box): bash scripts/dev/prof_infer.sh [tag] -> gpurun_out/prof_<tag>_{if3,if1}/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r03}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_if3 -- python3 $R/bench.py --no-cpu-baseline --no-extra > $R/gpurun_out/prof_${T}_if3.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_if1 -- python3 $R/bench.py --no-cpu-baseline --no-extra --inflight 1 > $R/gpurun_out/prof_${T}_if1.json 2>/dev/null
