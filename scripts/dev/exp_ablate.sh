#!/bin/bash
# What does each group of layers cost INSIDE the 3-batches-in-flight pipeline?  bench.py with that group's launches dropped
# from the plan (timing only; the results of such a run are garbage).  usage: bash scripts/dev/exp_ablate.sh [size]
S=${1:-320}
for a in "" "b.s3*.a,b.s4*.a,b.s5*.a" "b.stem,b.s1,b.down2,b.s2" "b.s5*.b,b.down5" "b.s4*.b,b.down4" "b.s3*.b,b.down3" "n.,h." "b.s3*.a" "b.stem" "b.s1" "b.s2"; do
  echo -n "ablate [$a]: "
  OD_ABLATE_OPS="$a" python bench.py --size $S --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step; net one-at-a-time', d['roofline']['network_ms_per_batch'])"
done
