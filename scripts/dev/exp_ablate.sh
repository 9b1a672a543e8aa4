#!/bin/bash
# What does each group of layers cost INSIDE the 3-batches-in-flight pipeline?  bench.py with that group's launches dropped
# from the plan (timing only; the results of such a run are garbage).  usage: bash scripts/dev/exp_ablate.sh [size]
# (Since the stage-3 1x1 layers ride in their producers' launches -- od_conv_desc.w2 -- "b.s3*.b,b.down3" drops them too,
#  and only stages 4-5 have 1x1 launches of their own.)
S=${1:-320}
for a in "" "b.s3*.b,b.down3" "b.s4*.b,b.down4" "b.s5*.b,b.down5" "b.s4*.a,b.s5*.a" "b.stem,b.s1,b.down2,b.s2" "b.stem" "b.s1" "b.down2" "b.s2" "n.lat5,n.lat4,n.out4,n.lat3,n.out3"; do
  echo -n "ablate [$a]: "
  OD_ALLOW_ABLATION=1 OD_ABLATE_OPS="$a" python bench.py --size $S --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step; net one-at-a-time', d['roofline']['network_ms_per_batch'])"
done
