#!/bin/bash
# usage: pmc_l2.sh <cfg> <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $R/gpurun_out/l2_$2_a -- python3 $R/scripts/dev/bench_conv.py --batch 128 --shapes w40 --cfgs $1 --nores > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/l2_$2_b -- python3 $R/scripts/dev/bench_conv.py --batch 128 --shapes w40 --cfgs $1 --nores > /dev/null 2>&1
