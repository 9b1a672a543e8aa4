#!/bin/bash
# training step (32 x 320^2) with the weight-gradient stream confined to a CU subset
run() { echo -n "[OD_TRAIN_WSTREAM_CUS=$1] "; OD_TRAIN_WSTREAM_CUS="$1" python bench.py --mode train --steps 8 --warmup 3 --reps 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms/step', d['ms_per_step_reps'])"; }
run ""
run "0:64"
run "0:96"
run "0:128"
run "0:64:4"
run "0:128:2"
run "0:32:8"
run "0:192"
run ""
