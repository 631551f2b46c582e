python -m pytest tests/test_gpu_parity.py tests/test_gpu_population.py -x -q 2>&1 | tail -2
pick='import json,sys
l=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], round(l["ms_per_step"],4), "ms/step", round(l["value"]), "GCUPS", {k: round(v,4) for k,v in l["kernel_ms"].items()})'
for rep in 1 2 3; do
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "$pick" run$rep
done
python bench.py --config 5 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('config5', l['ms_per_step'], l.get('kernel_ms'))"
