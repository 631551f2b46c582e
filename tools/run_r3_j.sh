pick='import json,sys
l=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], round(l["ms_per_step"],4), "ms/step", {k: round(v,4) for k,v in l["kernel_ms"].items()}, l["dp_launched_per_step"])'
python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "$pick" default
PLAT_NO_UNGAPPED=1 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "$pick" no_ungapped
PLAT_NO_UNGAPPED=1 PLAT_NO_EXACT=1 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "$pick" no_ungapped_no_exact
PLAT_SEED_DEBUG=256 python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "$pick" sweep_only
