run() {
  env "$@" python bench.py --config 4 --regions ${NREG:-3875} --steps 1 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$*', round(l['value']), 'win/s', 'T', round(l['timed_s'],3), 'Tcall', round(l['seconds_calls_mean_over_ranks'],3), 'gather_ms', round(l['record_gather']['ms'],1), 'host', round(l['host_seconds_per_region']*1e3,2), 'wait', round(l['device_wait_seconds_per_region']*1e3,2), 'src', round(l['source_seconds_per_region']*1e3,2), 'srcwait', round(l['worker_seconds_waiting_for_the_source_per_region']*1e3,2))
"
}
run PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run PLAT_CALLER_WORKERS=12 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run PLAT_CALLER_WORKERS=12 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=10
run PLAT_CALLER_WORKERS=10 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=10
run PLAT_CALLER_WORKERS=12 PLAT_CALLER_CHUNK=2 PLAT_CALLER_LOADERS=10
run PLAT_CALLER_WORKERS=14 PLAT_CALLER_CHUNK=3 PLAT_CALLER_LOADERS=10
NREG=7750 run PLAT_CALLER_WORKERS=12 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=10
