#!/bin/bash
# usage: pmc_conv.sh <cfg> <outdir>   (run on the GPU box from the repo root)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/$2_a -- python3 $R/scripts/dev/bench_conv.py --batch 128 --shapes w40 --cfgs $1 --nores > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/$2_b -- python3 $R/scripts/dev/bench_conv.py --batch 128 --shapes w40 --cfgs $1 --nores > /dev/null 2>&1
