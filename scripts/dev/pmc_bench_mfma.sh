#!/bin/bash
# MFMA / issue counters of every kernel of one bench.py run (one step at a time: --inflight 1, so that a dispatch's counters are its own)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/pmcm_a $R/gpurun_out/pmcm_b
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $R/gpurun_out/pmcm_a -- python3 $R/bench.py --steps 5 --warmup 2 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmcm_b -- python3 $R/bench.py --steps 5 --warmup 2 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
