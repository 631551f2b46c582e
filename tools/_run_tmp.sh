mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_stage_b.py tests/test_gpu_region_golden.py tests/test_gpu_caller.py -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && PLAT_CALLER_CHUNK=48 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -- python $GRAFT_REPO_ROOT/bench.py --config 4 --regions 1536 --steps 1 --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total GPU kernel time ms", tot/1e6, "per region us (2 passes x 1536 regions)", tot/1e3/(2*1536))
for r in rows[:12]:
    print("%-38s calls %5s avg %9.1f us  %5.2f%%" % (r['Name'].split('(')[0][:38], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
