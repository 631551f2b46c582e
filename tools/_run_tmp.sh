mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/pytest_gpu_tail.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/dp_mapping_a tools/ubench/dp_mapping_a.hip -Lplatypus_amd -lplat_mi355x && LD_LIBRARY_PATH=platypus_amd timeout 300 /tmp/dp_mapping_a 400000 150 | tee gpurun_out/mapping_a.json
( time timeout 900 python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','steps','gcups_all_dp','wgs_windows_per_sec','wgs_gcups','wgs_cpus_per_rank') if k in d})
print(d['config'].get('passes_per_step'), d['wgs'].get('scaling'), d['wgs'].get('stage_b'), d['wgs'].get('dp_per_launch'), d['wgs'].get('roofline',{}).get('kernel'), d['wgs'].get('streamed'))
PY
for ch in 64; do PLAT_CALLER_CHUNK=$ch timeout 300 python bench.py --config 4 --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $ch', d['value'], d['host_seconds_per_region'], d['device_wait_seconds_per_region'], d.get('gcups'), d.get('dp_per_launch'), d['scaling'])"; done
for ch in 32 128; do PLAT_CALLER_CHUNK3=$ch timeout 600 python bench.py --config 3 --steps 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end']; print('config3 chunk $ch tiles/s kernel', d['value'], 'end to end tiles/s', e['tiles_per_sec'], 'regions/s', e['regions_per_sec'], 'host', e['host_seconds_per_region'], 'wait', e['device_wait_seconds_per_region'])"; done
