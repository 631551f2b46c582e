mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_stage_b.py tests/test_gpu_region_golden.py tests/test_gpu_caller.py -q -x 2>&1 | tail -3
timeout 400 python tools/native_soak.py 200 2>&1 | tail -2
for cfg in "64 16" "32 16"; do set -- $cfg; ch=$1; wk=$2
PLAT_CALLER_CHUNK=$ch PLAT_CALLER_WORKERS=$wk timeout 300 python bench.py --config 4 --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $ch workers $wk  windows/s', round(d['value']), 'host', round(1e6*d['host_seconds_per_region'],1), 'wait', round(1e6*d['device_wait_seconds_per_region'],1), 'gcups', round(d.get('gcups',0),1), d['stage_b'], d.get('dp_per_launch'))"
done
echo "2 CPUs:"; PLAT_CALLER_CHUNK=32 PLAT_CALLER_WORKERS=3 PLAT_CALLER_LOADERS=2 taskset -c 0,1 timeout 300 python bench.py --config 4 --regions 1024 --steps 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  windows/s', d['value'], 'host', d['host_seconds_per_region'], 'wait', d['device_wait_seconds_per_region'])"
