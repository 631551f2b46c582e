run() {
  env "$@" python bench.py --config 4 --regions ${NREG:-1024} --steps 1 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$*', round(l['value']), 'win/s', round(l['regions_per_sec']), 'reg/s', 'T', round(l['timed_s'],3), 'Tcall', round(l['seconds_calls_mean_over_ranks'],3), 'host', round(l['host_seconds_per_region']*1e3,2), 'wait', round(l['device_wait_seconds_per_region']*1e3,2), 'src', round(l['source_seconds_per_region']*1e3,2), 'srcwait', round(l['worker_seconds_waiting_for_the_source_per_region']*1e3,2), 'h2d', round(l['h2d_gbytes_per_sec'],1), {k: round(v*1e3,2) for k,v in l['stage_seconds_per_region'].items()})
"
}
run PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8 PLAT_SYNC_SPIN=1
run PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=4
run PLAT_CALLER_WORKERS=12 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=4
run PLAT_CALLER_WORKERS=24 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=6
run PLAT_CALLER_WORKERS=32 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=6
run PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=8 PLAT_CALLER_LOADERS=6
run PLAT_CALLER_WORKERS=24 PLAT_CALLER_CHUNK=2 PLAT_CALLER_LOADERS=6
NREG=3875 run PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=6
