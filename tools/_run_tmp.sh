mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_stage_b.py tests/test_gpu_region_golden.py tests/test_gpu_caller.py -q -x 2>&1 | tail -12
for cfg in "48 14"; do set -- $cfg; ch=$1; wk=$2
PLAT_CALLER_CHUNK=$ch PLAT_CALLER_WORKERS=$wk PLAT_CALLER_TRACE=1 timeout 300 python bench.py --config 4 --steps 3 --no-cpu-baseline > gpurun_out/c4.json 2> gpurun_out/c4.err; grep "per region" gpurun_out/c4.err | tail -1; python -c "
import json,sys
d=json.loads(open('gpurun_out/c4.json').read().strip().splitlines()[-1]); print('  windows/s', d['value'], 'host', d['host_seconds_per_region'], 'wait', d['device_wait_seconds_per_region'], 'gcups', d.get('gcups'), d['stage_b'])"
done
echo "2 CPUs:"; PLAT_CALLER_CHUNK=32 PLAT_CALLER_WORKERS=3 PLAT_CALLER_LOADERS=2 taskset -c 0,1 timeout 300 python bench.py --config 4 --regions 1024 --steps 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  windows/s', d['value'], 'host', d['host_seconds_per_region'], 'wait', d['device_wait_seconds_per_region'], d['stage_b'])"
