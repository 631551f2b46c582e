# round 3: candidate kernels (8-byte compare scan, thread-per-read merge): parity, config-4 throughput, kernel stats
python -m pytest tests/test_gpu_caller.py tests/test_gpu_hostapi.py -x -q 2>&1 | tail -4
run() {
  env "$@" python bench.py --config 4 --regions ${NREG:-3875} --steps ${STEPS:-3} 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$*', round(l['value']), 'win/s', [round(x,3) for x in l['timed_s_runs']], 'host', round(l['host_seconds_per_region']*1e3,2), 'wait', round(l['device_wait_seconds_per_region']*1e3,2), 'src', round(l['source_seconds_per_region']*1e3,2), 'srcwait', round(l['worker_seconds_waiting_for_the_source_per_region']*1e3,2), {k: round(v*1e3,2) for k,v in l['stage_seconds_per_region'].items()})
"
}
run PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run PLAT_CALLER_WORKERS=12 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run PLAT_CALLER_WORKERS=14 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=6
run PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=8 PLAT_CALLER_LOADERS=8
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4prof -- python $GRAFT_REPO_ROOT/bench.py --config 4 --regions 256 --steps 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py /tmp/c4prof 2>/dev/null | head -30 || find /tmp/c4prof -name "*kernel_stats.csv" | head -1 | xargs head -30
