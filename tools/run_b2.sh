python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
for leg in 0 1; do PLAT_SEED_SHARE=$leg python bench.py --steps 200 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('share=$leg', round(l['value']), 'GCUPS', round(l['ms_per_step'],4), 'ms/step', {k: round(v,3) for k,v in l['kernel_ms'].items()}, l['dp_launched_per_step'])"; done
PLAT_NO_UNGAPPED=1 PLAT_NO_EXACT=1 python bench.py --steps 100 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('alldp', round(l['gcups_executed']), 'GCUPS', round(l['ms_per_step'],4), 'ms/step', {k: round(v,3) for k,v in l['kernel_ms'].items()}, l['dp_launched_per_step'])"
