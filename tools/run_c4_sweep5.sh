run() {
  env "$@" python bench.py --config 4 --regions 3875 --steps 3 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$*', round(l['value']), 'win/s', [round(x,3) for x in l['timed_s_runs']], 'host', round(l['host_seconds_per_region']*1e3,2), 'wait', round(l['device_wait_seconds_per_region']*1e3,2), 'src', round(l['source_seconds_per_region']*1e3,2), 'srcwait', round(l['worker_seconds_waiting_for_the_source_per_region']*1e3,2))
"
}
for rep in 1 2; do
run PLAT_CALLER_WORKERS=12 PLAT_CALLER_CHUNK=6 PLAT_CALLER_LOADERS=8
run PLAT_CALLER_WORKERS=12 PLAT_CALLER_CHUNK=8 PLAT_CALLER_LOADERS=8
run PLAT_CALLER_WORKERS=14 PLAT_CALLER_CHUNK=6 PLAT_CALLER_LOADERS=8
run PLAT_CALLER_WORKERS=12 PLAT_CALLER_CHUNK=6 PLAT_CALLER_LOADERS=6
run PLAT_CALLER_WORKERS=10 PLAT_CALLER_CHUNK=6 PLAT_CALLER_LOADERS=8
done
