#!/bin/bash
# rocprofv3 --kernel-trace --stats of the training step (two streams, and OD_TRAIN_WSTREAM=0)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r03}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_train -- python3 $R/bench.py --mode train --steps 10 --warmup 3 --reps 1 > $R/gpurun_out/prof_${T}_train.json 2>/dev/null
export OD_TRAIN_WSTREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_train1s -- python3 $R/bench.py --mode train --steps 10 --warmup 3 --reps 1 > $R/gpurun_out/prof_${T}_train1s.json 2>/dev/null
