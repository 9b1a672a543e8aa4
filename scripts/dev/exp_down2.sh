run() { echo -n "[$1] "; OD_TILE_CFG="$1" python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step; net', d['roofline']['network_ms_per_batch'])"; }
run ""
for c in 13 14 17 27 0 4 8 15; do run "b.down2=$c"; done
