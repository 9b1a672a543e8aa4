# specialised 64x64 (cfg 30, 31) vs the configs the plan uses, hot micro-benchmark and in the pipeline
python scripts/dev/bench_conv.py --shapes small --cfgs=-1,3,24,27,17,30,31 --nores --reps 20 2>/dev/null
run() { echo -n "[$1] "; OD_TILE_CFG="$1" python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step; net', d['roofline']['network_ms_per_batch'])"; }
run ""
run "b.s3*a=30"
run "b.s3*a=31"
run "b.s3*a=30,b.s5*a=30"
run "b.s3*a=31,b.s5*a=31,b.s4*a=31"
run "b.s3*a=30,b.s4*a=30,b.s5*a=30,n.lat5=30"
