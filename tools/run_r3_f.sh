python -m pytest tests/test_gpu_caller.py tests/test_gpu_hostapi.py -x -q 2>&1 | tail -3
run() {
  env "$@" python bench.py --config 4 --regions ${NREG:-3875} --steps 1 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$*', round(l['value']), 'win/s', [round(x,3) for x in l['timed_s_runs']], 'host', round(l['host_seconds_per_region']*1e3,2), 'wait', round(l['device_wait_seconds_per_region']*1e3,2), 'src', round(l['source_seconds_per_region']*1e3,2), 'srcwait', round(l['worker_seconds_waiting_for_the_source_per_region']*1e3,2), {k: round(v*1e3,2) for k,v in l['stage_seconds_per_region'].items()})
"
}
run PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run PLAT_CALLER_WORKERS=12 PLAT_CALLER_CHUNK=4 PLAT_CALLER_LOADERS=8
run PLAT_CALLER_WORKERS=16 PLAT_CALLER_CHUNK=8 PLAT_CALLER_LOADERS=8
bash tools/profile_round3.sh c4 > /dev/null 2>&1; f=$(find gpurun_out/r03p/stats_c4 -name "*kernel_stats.csv" | head -1); python - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ns", tot)
for r in rows[:12]: print(r["Name"][:50].ljust(50), r["Calls"].rjust(6), r["TotalDurationNs"].rjust(12), r["AverageNs"].rjust(10), r["Percentage"])
PY
