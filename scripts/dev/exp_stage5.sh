run() { echo -n "[$1 | $2] "; OD_TILE_CFG="$2" python bench.py $1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step; net', d['roofline']['network_ms_per_batch'])"; }
run "" ""
for c in 15 16 8 9 13 29 4; do run "" "b.s5*b=$c,b.down5=$c"; done
run "--inflight 1" ""
for c in 15 16 8 9; do run "--inflight 1" "b.s5*b=$c,b.down5=$c"; done
