run() {
  env "$@" python bench.py --config 4 --regions ${NREG:-3875} --steps 1 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$*', round(l['value']), 'win/s', round(l['regions_per_sec']), 'reg/s', 'T', round(l['timed_s'],3), 'host', round(l['host_seconds_per_region']*1e3,2), 'wait', round(l['device_wait_seconds_per_region']*1e3,2), 'src', round(l['source_seconds_per_region']*1e3,2), 'srcwait', round(l['worker_seconds_waiting_for_the_source_per_region']*1e3,2), {k: round(v*1e3,2) for k,v in l['stage_seconds_per_region'].items()})
"
}
run PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run GPU_MAX_HW_QUEUES=8 PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run GPU_MAX_HW_QUEUES=16 PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run GPU_MAX_HW_QUEUES=2 PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run GPU_MAX_HW_QUEUES=16 PLAT_CALLER_WORKERS=12 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run GPU_MAX_HW_QUEUES=16 PLAT_CALLER_WORKERS=20 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run GPU_MAX_HW_QUEUES=16 PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=8 PLAT_CALLER_LOADERS=10
