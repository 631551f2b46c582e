python -m pytest tests/test_gpu_caller.py -x -q 2>&1 | tail -5
python bench.py --config 3 --steps 10 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('config3', round(l['value']), 'tiles/s standalone;', json.dumps(l['end_to_end']))"
for w in "16 16" "16 32" "16 64" "12 32"; do set -- $w
PLAT_CALLER_WORKERS=$1 PLAT_CALLER_CHUNK3=$2 python - <<PY
import json
from tools import bench_other
r = bench_other.config3_end_to_end(0, 2000)
print("c3 e2e workers $1 chunk $2:", {k: (round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k not in ("text","gcups_note")})
PY
done
