#!/bin/bash
# usage: pmc_clk.sh <cfg> <dbg> <tag>   effective clock (GRBM_GUI_ACTIVE / 8 / duration) + wave cycles on a LONG dispatch
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
ARGS="$R/scripts/dev/bench_conv.py --batch 128 --shapes custom --custom 64,256,256,3,1 --cfgs $1 --nores --reps 10"
OD_CONV_DEBUG=$2 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/clk_$3 -- python3 $ARGS > /dev/null 2>&1
