run() { echo -n "[$*] "; env "$@" python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], d['ms_per_step_reps'])"; }
run A=1
run OD_INFLIGHT_CALIBRATE=0
run GPU_MAX_HW_QUEUES=8 OD_INFLIGHT_CALIBRATE=0
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=16 OD_INFLIGHT_CALIBRATE=0
run GPU_MAX_HW_QUEUES=2 OD_INFLIGHT_CALIBRATE=0
