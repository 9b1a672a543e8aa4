#!/bin/bash
# HBM-side traffic of every kernel of one bench.py run (separate --pmc passes, as the MI355X guide prescribes).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcb_fetch -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcb_write -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcb_stats -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/pmcb_bench.json 2>/dev/null
