#!/bin/bash
# usage: pmc_clock.sh <cfg> <dbg> <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
OD_CONV_DEBUG=$2 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/clk_$3 -- python3 $R/scripts/dev/bench_conv.py --batch 128 --shapes w40 --cfgs $1 --nores > /dev/null 2>&1
