#!/bin/bash
# batch-1 latency / throughput: eager plan vs hipGraph replay, 1 / 3 / 6 batches in flight
for s in 320 640; do for inf in 1 3 6; do for g in "" "--graph"; do
  echo -n "size $s inflight $inf $g: "
  python bench.py --size $s --batch 1 --inflight $inf $g --steps 200 --warmup 20 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
done; done; done
