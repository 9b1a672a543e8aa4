#!/bin/bash
# A/B of one environment knob on the default bench line, interleaved on ONE box: exp_ab.sh VAR A B [bench args]
V=$1; A=$2; B=$3; shift 3
ARGS="$@"
run() {
  export $V=$1
  python bench.py --no-cpu-baseline --no-extra $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V=$1', d['value'], 'img/s', d['ms_per_step'], 'ms/step; net one-at-a-time', d['roofline']['network_ms_per_batch'], 'dominant', d['roofline']['kernel'][:40], d['roofline']['frac'])"
}
for i in 1 2 3; do run $A; run $B; done
