#!/bin/bash
# usage: pmc_e8.sh <cfg> <dbg> <tag>   PMC passes on one exact-fit shape (M = 65536 = 256 tiles of 256, N = 256, K = 2304)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
ARGS="$R/scripts/dev/bench_conv.py --batch 16 --shapes custom --custom 64,256,256,3,1 --cfgs $1 --nores --reps 10"
OD_CONV_DEBUG=$2 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_$3_a -- python3 $ARGS > /dev/null 2>&1
OD_CONV_DEBUG=$2 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_$3_b -- python3 $ARGS > /dev/null 2>&1
OD_CONV_DEBUG=$2 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_$3_c -- python3 $ARGS > /dev/null 2>&1
