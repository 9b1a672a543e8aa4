#!/bin/bash
# Profiles of one bench.py run on MI355X: HBM-side traffic of every kernel (separate --pmc passes, as the MI355X guide
# prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass), then the kernel-trace --stats summary of the default command.
# usage (on the GPU box): bash scripts/dev/pmc_bench.sh [tag]     -> gpurun_out/pmcb_{fetch,write,stats}_<tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r02}
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcb_fetch_$T -- python3 $R/bench.py --steps 10 --warmup 3 --reps 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcb_write_$T -- python3 $R/bench.py --steps 10 --warmup 3 --reps 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcb_stats_$T -- python3 $R/bench.py --no-cpu-baseline --no-extra > $R/gpurun_out/pmcb_bench_$T.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcb_train_$T -- python3 $R/bench.py --mode train --steps 10 --warmup 3 --reps 1 > $R/gpurun_out/pmcb_train_$T.json 2>/dev/null
